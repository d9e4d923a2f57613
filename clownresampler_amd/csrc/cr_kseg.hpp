// cr_kseg.hpp - k_seg: a lane per stream SEGMENT, the 64 lanes of a wave on frames of EQUAL fraction; the polyphase row in scalar registers.
#ifndef CR_KSEG_HPP
#define CR_KSEG_HPP

#include "cr_device.hpp"

namespace
{

// ---------------------------------------------------------------------------------------------------------
// k_seg - long upsampling launches (cfg 3: stereo 8 -> 96 kHz, 8 lobes), round 5
// ---------------------------------------------------------------------------------------------------------
// Where k_up2's time goes (profiles/r05_kup2_lds_ablations.log, timing-only builds): the row a lane reads for every frame - four
// ds_read_b128, 64 bytes, 16 of a wave-step's ~34 LDS cycles, three VALU of addressing - is 18 % of the launch, and 64 lanes read
// only four or five DIFFERENT rows per step.  The reference's position is j * increment in 16.16 (clownresampler.h:1076-1078), so
// output frames j and j + S have the SAME fraction whenever S * increment is a multiple of 65536 (dual mono's observation, round 4):
// the same row, windows exactly D = S * increment / 65536 input frames apart.  Here the 64 lanes of a wave take the frames
// j, j + S, ..., j + 63 S - lane l walks segment l of a "super-block" of 64 S output frames - so at every step the whole wave needs
// ONE row, and:
//   * the row arrives by s_load_dwordx16 (from a 64-byte-per-row float image in global memory: L2 / scalar cache), a frame ahead,
//     and the taps take their weights as SCALAR operands of v_pk_fma_f32 (k_up2's FP32 round-toward-zero chain, cr_kup.hpp):
//     no row in LDS (66 KB free), no row address arithmetic in the vector unit, the fraction walk (g -= increment, wrap) is SALU;
//   * every lane is at the same place in its window: "the position advances" is a scalar branch, the window a register ring -
//     the frame body exists in TT rotations of the register names, so nothing is ever moved - fed one input frame per advance;
//   * no lane is ever idle inside a tile (k_up2: 60 of 64 positions per wave-tile), no predicated frames, no per-position
//     bookkeeping (first_frame_of, start / count / extra);
//   * a lane's frames are consecutive in ITS segment: they are staged 16 at a time (one 128-byte line per lane, rows 136 bytes
//     apart: ds_write_b64 free of bank conflicts - k_up2's staging writes were 4-way conflicted at 12 frames per position) and
//     leave as 16 buffer stores of four whole lines each.
// What it needs: S a multiple of 65536 / gcd(increment, 65536) (65536 frames for an odd increment such as cfg 3's 5461), a launch
// of many super-blocks (the lanes of the last, partial one idle: the host takes k_seg when that waste is small), fixed slot signs
// (NEGMASK, as k_up2), a stream below 4 GiB either side of a super-block (32-bit buffer offsets).
// A tile = K consecutive frames (a multiple of 16) of each of the 64 segments of one super-block; tiles are drawn as tickets.
// ---------------------------------------------------------------------------------------------------------
// A lane's frames are staged CHUNK at a time (CHUNK * 8 bytes per lane: a 128-byte line, or half of one) in rows CHUNK * 8 + 8 bytes apart
// (16 neighbouring lanes on 32 different banks either way), and leave as CHUNK buffer stores of 64 / CHUNK whole pieces each.
constexpr unsigned seg_lane_stride(unsigned chunk) { return chunk * 8u + 8u; }
// The input frames a lane will advance onto wait in a RING in LDS, filled by LDS-DMA four frames per lane at a time (a "group":
// buffer_load_dwordx4 ... lds, every lane from its own place in the stream, 1 KiB per group: [group][lane][4 frames]).
constexpr unsigned seg_wave_bytes(unsigned chunk, unsigned groups) { return 64u * seg_lane_stride(chunk) + groups * 1024u; }

// ABL (timing-only diagnostic instances, results WRONG): 1 = one row per tile (no scalar loads in the frame loop), 2 = no global stores,
// 3 = both.
template <int TT, unsigned NEGMASK, int WAVES, int NT, unsigned CHUNK, unsigned NG, int ABL = 0>
__global__ __launch_bounds__(WAVES * 64) void k_seg(const crhip_seg_launch a)
{
	static_assert(TT == 15 && NEGMASK == 0x2A55u, "the frame body is written out for 15 slots, slots 0, 2, 4, 6, 9, 11, 13 negative");
	static_assert((CHUNK == 16u || CHUNK == 8u) && (NG & (NG - 1u)) == 0 && NG >= 4u, "a line or half a line per lane and chunk; a power of two of ring groups, the first window's four at least");
	typedef float f32x2 __attribute__((ext_vector_type(2)));
	typedef float f32x16 __attribute__((ext_vector_type(16)));
	constexpr unsigned LANE_STRIDE = seg_lane_stride(CHUNK);
	constexpr unsigned WAVE_BYTES = seg_wave_bytes(CHUNK, NG);
	constexpr unsigned PER_STORE = 64u / CHUNK;   // segments one store instruction covers (CHUNK lanes, 8 bytes each, per segment)

	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const unsigned lane = threadIdx.x & 63u;
	const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	unsigned char *my_stage = smem + wave * WAVE_BYTES;
	unsigned char *my_ring = my_stage + 64u * LANE_STRIDE;
	unsigned *waves_done = reinterpret_cast<unsigned *>(smem + WAVES * WAVE_BYTES);

	if (threadIdx.x == 0)
		*waves_done = 0;
	__syncthreads();

	// diagnostic instance only (ABL == 6): where wave 0 of every workgroup spends its shader cycles - [0] a tile's first window (requests,
	// wait, conversions), [1] the wait for the scalar row (and the LDS) at the head of every frame, [2] the frames' arithmetic, [3] copy-outs,
	// [4] the counted wait + ring requests before a chunk, [5] position advances, [6] frames, [7] tiles
	unsigned long long phase[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_mark = 0;
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

	// ---- tiles as tickets (k_up2's scheme: the first by global wave number, the rest from 32 counter lanes) ----
	const uint64_t n_tiles = a.n_tiles;
	const uint64_t global_wave = (uint64_t)wave * gridDim.x + blockIdx.x;
	const uint64_t global_waves = (uint64_t)gridDim.x * WAVES;
	const unsigned LANES = global_waves < 32u ? (unsigned)global_waves : 32u;
	const unsigned lane_id = (unsigned)(global_wave % LANES);
	// Which tiles a sequence takes.  Plain: lane_id, lane_id + 32, ... - neighbouring tiles, whose lanes read neighbouring BYTES of the same
	// 64 input lines (a tile advances a lane ~43 bytes at 12x), then run on waves of eight different XCDs, and every XCD's L2 fetches the
	// line for itself: 87.5 MiB fetched for 18.3 MiB of input on cfg 3 (profiles/r05_cfg3_pmc_summary.txt).  With a.xcd_run = G (a multiple
	// of 4; the grid a multiple of 8, so that a workgroup's XCD - its number mod 8, as the dispatcher deals them - is its waves' lane_id
	// mod 8): the tiles go out in rounds of 8 G, XCD x takes the G CONSECUTIVE tiles [G x, G x + G) of each round, its four sequences
	// every fourth of them - the runs of one round are in flight together on one XCD, and a line is fetched once per run.
	const unsigned G = (LANES == 32u && (gridDim.x & 7u) == 0u) ? a.xcd_run : 0u;
	const unsigned xcd = lane_id & 7u, sub = lane_id >> 3, per = G / 4u;   // (sub: 0 ... 3)
	auto tile_of = [&](uint64_t k) -> uint64_t {
		if (G == 0u)
			return lane_id + (uint64_t)LANES * k;
		return (k / per) * (8ull * G) + (uint64_t)G * xcd + 4ull * (k % per) + sub;
	};
	uint64_t lane_tiles;   // tiles of this sequence: the k with tile_of(k) < n_tiles (tile_of ascends)
	if (G == 0u)
		lane_tiles = n_tiles > lane_id ? (n_tiles - lane_id + LANES - 1u) / LANES : 0;
	else
	{
		const uint64_t rem = n_tiles % (8ull * G), first = (uint64_t)G * xcd + sub;
		const uint64_t partial = rem > first ? (rem - first + 3u) / 4u : 0;
		lane_tiles = (n_tiles / (8ull * G)) * per + (partial < per ? partial : per);
	}
	const unsigned lane_waves = (unsigned)((global_waves - lane_id + LANES - 1u) / LANES);
	unsigned *lane_counter = a.d_tickets + lane_id * 32u;
	auto draw_resolve = [&](unsigned got) -> uint64_t {
		const uint64_t k = (uint64_t)lane_waves + (unsigned)__builtin_amdgcn_readfirstlane((int)got);
		return k < lane_tiles ? tile_of(k) : ~0ull;
	};
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

	// ---- launch constants (wave-uniform) ----
	const unsigned increment = __builtin_amdgcn_readfirstlane(a.increment);
	const unsigned K = __builtin_amdgcn_readfirstlane(a.tile_frames);
	const uint64_t S = a.seg_frames;
	const unsigned tiles_per_seg = __builtin_amdgcn_readfirstlane(a.tiles_per_seg);
	// input frames a chunk can advance over, at most: what the ring must hold ahead of the wave (see top_up)
	const unsigned chunk_advances = (65535u + CHUNK * increment) >> 16;
	// (constant address space: a uniform load from it is an s_load whatever the kernel stores elsewhere)
	const __attribute__((address_space(4))) f32x16 *rows = (const __attribute__((address_space(4))) f32x16 *)(uintptr_t)a.d_rows;
	const uint64_t in_base = reinterpret_cast<uint64_t>(a.d_in);
	const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
	    reinterpret_cast<void *>(((uint64_t)__builtin_amdgcn_readfirstlane((int)(unsigned)(in_base >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)in_base)), 0,
	    (int)__builtin_amdgcn_readfirstlane((int)(unsigned)(a.in_valid_bytes > 0xFFFFFFFCull ? 0xFFFFFFFCull : a.in_valid_bytes)), 0x00020000);
	// lane l's window lies l * D input frames behind lane 0's (4 bytes per stereo frame; the host keeps everything below 2^32)
	const unsigned lane_in_bytes = lane * (unsigned)a.seg_in_frames * 4u;

	// {max(v, 0), max(-v, 0)} of both channels of a packed input frame, as floats: the sample's positive and negative part, both as magnitudes
	auto convert = [&](int packed, f32x2 &left, f32x2 &right) {
		float v0, v1;
		asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(v0) : "v"(packed));
		asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(v1) : "v"(packed));
		asm("v_max_f32_e64 %0, %1, 0" : "=v"(left.x) : "v"(v0));
		asm("v_max_f32_e64 %0, -%1, 0" : "=v"(left.y) : "v"(v0));
		asm("v_max_f32_e64 %0, %1, 0" : "=v"(right.x) : "v"(v1));
		asm("v_max_f32_e64 %0, -%1, 0" : "=v"(right.y) : "v"(v1));
	};

	// Where a frame's two chains start: 2^23, ulp 1 (cr_kup.hpp, FCHAIN).  Unlike k_up2, BOTH chains count upwards: one sums the products that
	// are positive, the other the MAGNITUDES of the negative ones (a sample's negative part is kept as a magnitude, a slot with negative weights
	// - staged as magnitudes too - takes the pair swapped: negative part x |weight| to the first chain, positive part x |weight| to the second).
	// Round-toward-zero truncates either sum as C truncates the products (clownresampler.h:1020 via :625), and the frame's sum is the difference
	// of the two accumulators' BITS (same exponent): one subtraction per channel, no sign to patch.
	f32x2 chain_base;
	chain_base.x = 8388608.0f;
	chain_base.y = 8388608.0f;
	asm volatile("" : "+v"(chain_base));
	int thirty_one = 31;
	asm volatile("" : "+v"(thirty_one));
	int zero_c = 0, zero_d = 0;
	asm volatile("" : "+{v117}"(zero_c), "+{v119}"(zero_d));   // the high halves of the normalisation's pinned addend pairs (see k_up2)

	// staging: this lane's row, and where it reads for the copy-out (store i of a chunk: segments 4 i ... 4 i + 3, 16 lanes each)
	// a lane's staging row: 8 bytes that take the write of "no frame yet", then its CHUNK frames
	const unsigned stage_row = (unsigned)(uintptr_t)my_stage + lane * LANE_STRIDE;
	// (store i of a chunk: segments PER_STORE * i ... + PER_STORE - 1, CHUNK lanes each)
	const unsigned copy_from = (unsigned)(uintptr_t)my_stage + (lane / CHUNK) * LANE_STRIDE + 8u + (lane % CHUNK) * 8u;
	const unsigned ring_at = (unsigned)(uintptr_t)my_ring + lane * 16u;   // this lane's four frames of group 0

	// a wave's first tile: its rank among the waves of its sequence
	if (global_wave / LANES >= lane_tiles)
	{
		retire();
		return;
	}
	uint64_t tile = tile_of(global_wave / LANES);

	for (;;)
	{
		// the ticket for the tile after this one: its round trip runs under the tile
		// (the counter's address rebuilt from scalar halves: with this kernel's scalar register pressure hipcc otherwise parks the pointer in a VGPR pair,
		// which the scalar atomic cannot take)
		const uint64_t counter_bits = reinterpret_cast<uint64_t>(lane_counter);
		unsigned *const counter = reinterpret_cast<unsigned *>(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(counter_bits >> 32)) << 32)
		                                                       | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)counter_bits));
		const unsigned ticket = draw_ticket_begin(counter);
		mark(-1);

		// ---- where the tile is ----
		// (everything below is wave-uniform by construction: say so, or hipcc carries the tile's 64-bit bookkeeping through the vector unit
		// and wraps every store in a waterfall loop; the host keeps the tile count below 2^32)
		const unsigned tile_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)tile);
		const unsigned block = (unsigned)__builtin_amdgcn_readfirstlane((int)(tile_lo / tiles_per_seg));
		const unsigned t = (unsigned)__builtin_amdgcn_readfirstlane((int)(tile_lo - block * tiles_per_seg));
		const uint64_t first = (uint64_t)block * 64u * S + (uint64_t)t * K;           // lane 0's first frame; lane l's: first + l * S
		const uint64_t pos = a.pos0 + first * (uint64_t)increment;           // 16.16
		const uint64_t position = (pos >> 16) + a.first_slot;                // input frame that slot 0 of lane 0's first frame multiplies
		// the fraction of the wave's current frame in the TOP half of a register: adding the increment (shifted likewise) carries exactly
		// when the position advances, and nothing has to be wrapped back (scalar: 2 instructions per frame)
		unsigned F = (unsigned)(pos & 0xFFFFu) << 16;
		const unsigned INC = increment << 16;                                // (increment < 65536: upsampling)

		// ---- the input side: entry n of the tile is input frame position + n of every lane's own stretch of the stream (beyond the caller's
		//      buffer: zeros); group q = entries 4 q ... 4 q + 3 sits in ring slot q mod NG.  The first window is entries 0 ... TT - 1. ----
		const unsigned in_at = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(position * 4u)) + lane_in_bytes;
		unsigned loaded = 0;   // groups requested so far (wave-uniform)
		auto request_group = [&]() {
			__builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (__attribute__((address_space(3))) void *)(my_ring + (loaded & (NG - 1u)) * 1024u), 16,
			                                         (int)(in_at + loaded * 16u), 0, 0, 0);
			++loaded;
		};
		// Before a chunk: everything the wave can advance onto until the end of the NEXT chunk has been requested (what it needs in THIS
		// chunk was requested a chunk ago, ahead of that chunk's stores: the counted wait in front of the chunk covers it).
		auto top_up = [&](unsigned advanced) {
			const unsigned need = (unsigned)TT + advanced + 2u * chunk_advances + 1u;   // entries
			while (loaded * 4u < need)
				request_group();
		};
		// The first window comes straight into registers (four 16-byte loads per lane: entries 0 ... 15, the last one unused); the ring
		// starts with group 3 - entry TT = 15, the frame the first advance brings in, is its last - and everything lands under ONE wait.
		static_assert(TT == 15, "the ring starts at group 3: entry 15 is the first advance's");
		int raw[16];
#pragma unroll
		for (int q = 0; q < 4; ++q)
		{
			const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)(in_at + 16u * q), 0, 0);
			raw[4 * q] = v.x;
			raw[4 * q + 1] = v.y;
			raw[4 * q + 2] = v.z;
			raw[4 * q + 3] = v.w;
		}
		loaded = 3u;
		top_up(0);
		// (the previous tile's last stores are older than these requests: all of it has to land - once per tile)
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		f32x2 P[TT][2];   // slot s of rotation R is P[(s + R) % TT]
#pragma unroll
		for (int s = 0; s < TT; ++s)
			convert(raw[s], P[s][0], P[s][1]);
		mark(0);
		if constexpr (ABL == 6)
			phase[7] += 1;

		// ---- the output side of the tile: 16 descriptors, one per store of a chunk (segments 4 i ... 4 i + 3), re-based per chunk ----
		const uint64_t out_first = reinterpret_cast<uint64_t>(a.d_out) + first * 8u;
		const uint64_t out_end = reinterpret_cast<uint64_t>(a.d_out) + a.n_out * 8u;
		const unsigned store_at = (unsigned)((lane / CHUNK) * S * 8u) + (lane % CHUNK) * 8u;   // (< 2^32: the host)

		unsigned frames_left = K;
		{
			const uint64_t seg_left = S - (uint64_t)t * K;   // (the last tile of a segment may be shorter: a multiple of 16 all the same)
			if (seg_left < frames_left)
				frames_left = (unsigned)seg_left;
		}
		frames_left = __builtin_amdgcn_readfirstlane(frames_left);
		unsigned in_chunk = CHUNK;   // frames until the chunk is copied out (wave-uniform)
		unsigned chunk = 0;         // chunks copied out
		unsigned stage_at = stage_row;
		unsigned advance = 0;       // input frames advanced onto so far
		int next_frame = 0;         // the frame the next advance brings in (read from LDS a position ahead)

		// The row of a frame: (65536 - fraction) >> 6 (pure-upsampling row index), 64 bytes per row: s_load_dwordx16, a frame ahead.
		auto load_row = [&](unsigned fraction_hi) -> f32x16 {
			const unsigned off = (0x10000u - (fraction_hi >> 16)) & 0x1FFC0u;
			return *reinterpret_cast<const __attribute__((address_space(4))) f32x16 *>(reinterpret_cast<const __attribute__((address_space(4))) unsigned char *>(rows) + off);
		};
		f32x16 w_a = load_row(F);

		// bytes from the tile's first frame (segment 0) to the end of the stream, as far as 32 bits see (the host keeps a whole
		// super-block - 64 S frames of 8 bytes - below 2^32, so whatever is clamped away here lies beyond every store of the tile)
		const uint64_t room64 = out_end > out_first ? out_end - out_first : 0;
		const unsigned room = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(room64 > 0xFFFFFFFCull ? 0xFFFFFFFCull : room64));
		const unsigned store_step = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(S * 8u * PER_STORE));   // the next store's segments

		// A frame's result is written to its staging slot DURING THE NEXT frame (between that frame's two blocks of taps): the wait for the
		// scalar row load at the head of every frame is lgkmcnt(0) - scalar loads return out of order, nothing less will do - and it would
		// otherwise sit out the LDS write issued just in front of it, every frame.  `pending` is the frame waiting to be written and
		// stage_at where it goes: the slot of the wave's LAST frame - before the first frame of a chunk the 8 bytes in front of the row.
		i32x2 pending;
		pending.x = 0;
		pending.y = 0;
		auto write_pending = [&]() {
			asm volatile("ds_write_b64 %0, %1" ::"v"(stage_at), "v"(pending) : "memory");
		};

		auto copy_out = [&]() {
			write_pending();   // the chunk's last frame
			// the staged frames of the other lanes: same wave, LDS operations complete in order
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			// every read first, then the stores (one wait for the LDS instead of one per store)
			i32x2 v[CHUNK];
#pragma unroll
			for (int i = 0; i < (int)CHUNK; ++i)
				v[i] = *reinterpret_cast<const __attribute__((address_space(3))) i32x2 *>((uintptr_t)(copy_from + (unsigned)i * PER_STORE * LANE_STRIDE));
			// ONE descriptor per chunk: from segment 0's piece of this chunk to the end of the stream; store i goes PER_STORE segments further
			// on each time through its SCALAR offset, which the range check of a raw buffer on gfx9 / gfx950 takes into account (a piece,
			// or a segment, beyond the stream's end is dropped: tests/test_gpu_parity.py::test_segment_kernel_bit_exact holds a guard
			// over a whole block of segments behind the output)
			const unsigned chunk_off = chunk * (CHUNK * 8u);
			const uint64_t at = out_first + chunk_off;
			unsigned left;
			asm("s_sub_u32 %0, %1, %2\n\ts_cselect_b32 %0, 0, %0" : "=&s"(left) : "s"(room), "s"(chunk_off) : "scc");
			const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
			    reinterpret_cast<void *>(((uint64_t)__builtin_amdgcn_readfirstlane((int)(unsigned)(at >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)at)), 0,
			    (int)left, 0x00020000);
			unsigned soff = 0, step = store_step;
			asm volatile("" : "+s"(step));   // (one running sum, an add per store: hipcc otherwise keeps all sixteen multiples in scalar registers across the tile)
#pragma unroll
			for (int i = 0; i < (int)CHUNK; ++i)
			{
				if constexpr (ABL == 2 || ABL == 3)
					asm volatile("" ::"v"(v[i]), "s"(soff));
				else
					__builtin_amdgcn_raw_buffer_store_b64(v[i], rsrc, (int)store_at, (int)soff, NT ? 2 : 0);
				soff += step;
			}
			__builtin_amdgcn_wave_barrier();
			++chunk;
			stage_at = stage_row;
		};

		// one frame of every lane with the window in rotation R; the weights: the scalar row `w`
		auto frame = [&](auto r_tag, const f32x16 &w) {
			constexpr int R = decltype(r_tag)::value;
			auto wpair = [&](int k) {
				f32x2 wp;
				wp.x = w[2 * k];
				wp.y = w[2 * k + 1];
				return wp;
			};
#define CR_SEL_EP " op_sel_hi:[1,0,1]\n\t"
#define CR_SEL_OP " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
#define CR_SEL_EN " op_sel:[1,0,0] op_sel_hi:[0,0,1]\n\t"
#define CR_SEL_ON " op_sel:[1,1,0] op_sel_hi:[0,1,1]\n\t"
#define CR_T(acc, p, wq, SEL) "v_pk_fma_f32 %" #acc ", %" #p ", %" #wq ", %" #acc SEL
#define CR_F(acc, p, wq, base, SEL) "v_pk_fma_f32 %" #acc ", %" #p ", %" #wq ", %" #base SEL
#define CR_P(s, c) P[((s) + R) % TT][c]
			f32x2 a0, a1;
			// slots 0 (-), 1 (+), 2 (-), 3 (+), 4 (-), 5 (+), 6 (-), 7 (+): cr_kup.hpp, the same strings with scalar weight pairs
			asm volatile(CR_F(0, 2, 18, 22, CR_SEL_EN) CR_F(1, 10, 18, 22, CR_SEL_EN) CR_T(0, 3, 18, CR_SEL_OP) CR_T(1, 11, 18, CR_SEL_OP)
			             CR_T(0, 4, 19, CR_SEL_EN) CR_T(1, 12, 19, CR_SEL_EN) CR_T(0, 5, 19, CR_SEL_OP) CR_T(1, 13, 19, CR_SEL_OP)
			             CR_T(0, 6, 20, CR_SEL_EN) CR_T(1, 14, 20, CR_SEL_EN) CR_T(0, 7, 20, CR_SEL_OP) CR_T(1, 15, 20, CR_SEL_OP)
			             CR_T(0, 8, 21, CR_SEL_EN) CR_T(1, 16, 21, CR_SEL_EN) CR_T(0, 9, 21, CR_SEL_OP) CR_T(1, 17, 21, CR_SEL_OP)
			             : "=&v"(a0), "=&v"(a1)
			             : "v"(CR_P(0, 0)), "v"(CR_P(1, 0)), "v"(CR_P(2, 0)), "v"(CR_P(3, 0)), "v"(CR_P(4, 0)), "v"(CR_P(5, 0)), "v"(CR_P(6, 0)), "v"(CR_P(7, 0)),
			               "v"(CR_P(0, 1)), "v"(CR_P(1, 1)), "v"(CR_P(2, 1)), "v"(CR_P(3, 1)), "v"(CR_P(4, 1)), "v"(CR_P(5, 1)), "v"(CR_P(6, 1)), "v"(CR_P(7, 1)),
			               "s"(wpair(0)), "s"(wpair(1)), "s"(wpair(2)), "s"(wpair(3)), "v"(chain_base));
			write_pending();   // the previous frame's result: its LDS write has the rest of this frame to complete
			// slots 8 (+), 9 (-), 10 (+), 11 (-), 12 (+), 13 (-), 14 (+)
			asm volatile(CR_T(0, 2, 16, CR_SEL_EP) CR_T(1, 9, 16, CR_SEL_EP) CR_T(0, 3, 16, CR_SEL_ON) CR_T(1, 10, 16, CR_SEL_ON)
			             CR_T(0, 4, 17, CR_SEL_EP) CR_T(1, 11, 17, CR_SEL_EP) CR_T(0, 5, 17, CR_SEL_ON) CR_T(1, 12, 17, CR_SEL_ON)
			             CR_T(0, 6, 18, CR_SEL_EP) CR_T(1, 13, 18, CR_SEL_EP) CR_T(0, 7, 18, CR_SEL_ON) CR_T(1, 14, 18, CR_SEL_ON)
			             CR_T(0, 8, 19, CR_SEL_EP) CR_T(1, 15, 19, CR_SEL_EP)
			             : "+v"(a0), "+v"(a1)
			             : "v"(CR_P(8, 0)), "v"(CR_P(9, 0)), "v"(CR_P(10, 0)), "v"(CR_P(11, 0)), "v"(CR_P(12, 0)), "v"(CR_P(13, 0)), "v"(CR_P(14, 0)),
			               "v"(CR_P(8, 1)), "v"(CR_P(9, 1)), "v"(CR_P(10, 1)), "v"(CR_P(11, 1)), "v"(CR_P(12, 1)), "v"(CR_P(13, 1)), "v"(CR_P(14, 1)),
			               "s"(wpair(4)), "s"(wpair(5)), "s"(wpair(6)), "s"(wpair(7)));
#undef CR_P
#undef CR_F
#undef CR_T
#undef CR_SEL_ON
#undef CR_SEL_EN
#undef CR_SEL_OP
#undef CR_SEL_EP
			// first chain = 2^23 + p, second = 2^23 + q: the frame's sum is p - q
			const int hi0 = (int)(__float_as_uint(a0.x) - __float_as_uint(a0.y));
			const int hi1 = (int)(__float_as_uint(a1.x) - __float_as_uint(a1.y));
			// (acc * reciprocal) / 32768 toward zero (clownresampler.h:1033), the row's last entry being 2 * reciprocal: k_up2's form
			int out0, out1;
			const int reciprocal2 = (int)__float_as_uint(w[TT]);
			asm("v_ashrrev_i32_sdwa v116, %4, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
			    "v_ashrrev_i32_sdwa v118, %4, %3 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
			    "v_mad_i64_i32 v[122:123], vcc, %2, %5, v[116:117]\n\t"
			    "v_mad_i64_i32 v[126:127], vcc, %3, %5, v[118:119]\n\t"
			    "v_alignbit_b32 %0, v123, v122, 16\n\t"
			    "v_alignbit_b32 %1, v127, v126, 16"
			    : "=&v"(out0), "=&v"(out1)
			    : "v"(hi0), "v"(hi1), "v"(thirty_one), "s"(reciprocal2), "{v117}"(zero_c), "{v119}"(zero_d)
			    : "vcc", "v116", "v118", "v122", "v123", "v126", "v127");
			pending.x = out0;
			pending.y = out1;
		};

		// FP32 rounding: toward zero, for the chains - and the mode the wave came with put back behind the tile (not a
		// hard-coded 0).  Nothing between the two but the kernel's own asm arithmetic and integer / scalar code.
		unsigned fp_mode;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE, 0, 2)\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3" : "=s"(fp_mode));

		// the frame the first advance brings in
		// entry TT + k is the frame advance k + 1 brings in: [group mod NG][lane][entry mod 4]
		auto entry_at = [&](unsigned e) { return ring_at + ((e >> 2) & (NG - 1u)) * 1024u + (e & 3u) * 4u; };
		next_frame = *reinterpret_cast<const __attribute__((address_space(3))) int *>((uintptr_t)entry_at((unsigned)TT));

		while (frames_left != 0)
		{
			static_for<TT>([&](auto r_tag) {
				constexpr int R = decltype(r_tag)::value;
				if (frames_left != 0)
				{
					// the frames at this position: until the fraction wraps (or the tile ends)
					unsigned carry;
					do
					{
						// the next frame's row, requested before this frame's arithmetic (F + INC and its carry as two scalar instructions:
						// written in C++ hipcc does the add in the vector unit for its carry-out)
						unsigned next;
						asm("s_add_u32 %0, %2, %3\n\ts_cselect_b32 %1, 1, 0" : "=&s"(next), "=s"(carry) : "s"(F), "s"(INC) : "scc");
						if constexpr (ABL == 6)
						{
							mark(-1);
							asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
							asm volatile("" : "+s"(w_a));
							mark(1);
						}
						f32x16 w_next = w_a;
						if constexpr (ABL != 1 && ABL != 3)
							w_next = load_row(next);
						frame(r_tag, w_a);
						// (the rows TWO frames ahead - a third set of scalar registers - measured slower: 125 against 122 us, spills;
						// profiles/r05_kseg_ab2.log)
						w_a = w_next;
						if constexpr (ABL == 6)
						{
							mark(2);
							phase[6] += 1;
						}
						F = next;
						stage_at += 8u;
						if (--in_chunk == 0)
						{
							mark(-1);
							copy_out();
							mark(3);
							in_chunk = CHUNK;
							frames_left -= CHUNK;
							if (frames_left == 0)
								carry = 2u;   // (the tile ends: leave the loop, no advance - asked once per chunk, not once per frame)
							else
							{
								// the groups requested before this chunk have landed once only its own stores are outstanding (vmcnt counts
								// loads and stores alike, in order); then the requests the chunk after the next will need
								if constexpr (ABL == 2 || ABL == 3)
									asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
								else
									asm volatile("s_waitcnt vmcnt(%0)" ::"n"((int)CHUNK) : "memory");
								top_up(advance + carry);
								mark(4);
							}
						}
					} while (carry == 0);
					if (carry == 1u)
					{
						mark(-1);
						// the position advances: the oldest slot's registers take the new frame, the names rotate by one
						convert(next_frame, P[R % TT][0], P[R % TT][1]);
						++advance;
						next_frame = *reinterpret_cast<const __attribute__((address_space(3))) int *>((uintptr_t)entry_at((unsigned)TT + advance));
						mark(5);
					}
				}
			});
		}

		asm volatile("s_setreg_b32 hwreg(HW_REG_MODE, 0, 2), %0" ::"s"(fp_mode));

		const uint64_t next = draw_resolve(draw_ticket_end(ticket));
		if (next == ~0ull)
			break;
		tile = next;
	}

	if constexpr (ABL == 6)
	{
		if (lane == 0 && wave == 0 && a.debug_stamps != nullptr && blockIdx.x < 256u)
			for (int q = 0; q < 8; ++q)
				a.debug_stamps[8u * blockIdx.x + q] = phase[q];
	}
	retire();
}

} // namespace

#endif // CR_KSEG_HPP
