// cr_kpoly.hpp - k_poly: polyphase rows in LDS, workgroup tiles behind LDS-DMA (see cr_kernels.hip for the overview).
#ifndef CR_KPOLY_HPP
#define CR_KPOLY_HPP

#include "cr_device.hpp"

namespace
{

// ---------------------------------------------------------------------------------------------------------
// k_poly
// ---------------------------------------------------------------------------------------------------------
// CH        channels (compile time)
// TT        slots when > 0 (fully unrolled, weights in registers); 0 = run-time slot count
// MODE      row-index formula
// NORM      final normalisation form (CRHIP_NORM_*)
// NTHREADS  workgroup size
// NV        16-byte input vectors each thread moves per tile (LDS tile buffer = NV * 16 * NTHREADS bytes)
// ASM       1 = SDWA tap arithmetic, 0 = what the compiler makes of the C expression
// U         output frames a lane works on at once (independent instruction streams to cover LDS latency)
// SWZ       1 = the LDS image of the rows is swizzled (a.swizzle), 0 = plain (a.swizzle must be 0)
// ABL       0 in every shipped instance.  Timing-only ablations (WRONG results, reachable only through the debug
//           launch flag of tools/): 1 = no output stores, 2 = no input DMA, 3 = neither, 4 = DMA + stores but no arithmetic;
//           + 16 = every lane reads the window of the tile's first frame (window reads without bank conflicts), + 32 = every lane
//           reads row 0 (row reads without) - the pipelined frames of the specialised instances only
// OUT16     1 = clamp to +-0x7FFF and store int16 (opt-in extension), 0 = the reference's unclamped int32
// NT        1 = non-temporal output stores
// SPLIT     lanes per frame: CH is then the channels of ONE lane and a frame has CH * SPLIT channels (8-channel
//           frames as two lanes of 4: every store instruction of a wave is one contiguous 1 KiB)
// PH        1 = the frame has CH * SPLIT - 1 channels (odd totals above 8, SPLIT == 2): the second lane's last channel is a
//           PHANTOM - it multiplies whatever follows the frame in the window and its result is never stored
// DUAL      1 = DUAL MONO (round 4): a stereo instance run on a MONO stream.  Output frames j and j + H of one stream have the same
//           fractional position whenever H * increment is a multiple of 65536 - hence the same row - so they can be the two
//           "channels" of one lane: one row index, one row read, one reciprocal for two frames, and the frame body is the stereo
//           instance's, unchanged.  Per tile the two mono input windows (H * increment / 65536 input frames apart) are fetched by
//           the two halves of the workgroup into the two halves of the DMA buffer, interleaved into a stereo tile by an LDS -> LDS
//           pass (four frames per thread: four funnel shifts, four byte permutes, one 16-byte write), and the results leave as two 4-byte stores H frames
//           apart.  crhip_poly_launch.dual* carry H, how many second frames exist, and the input offset; n_out counts pairs.
template <int CH, int TT, int MODE, int NORM, int NTHREADS, int NV, int ASM, int U, int SWZ, int ABL = 0, int OUT16 = 0, int NT = 0, int SPLIT = 1, int PH = 0, int DUAL = 0, int PADT = 0>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_num_sgpr(CRHIP_SGPR_BUDGET))) void k_poly(const crhip_poly_launch a)
{
	static_assert(!DUAL || (CH == 2 && SPLIT == 1 && PH == 0 && OUT16 == 0 && TT > 0 && ABL == 0 && (NTHREADS / 2) % 64 == 0), "dual mono: a stereo instance, int32 output");
	constexpr unsigned CHT = CH * SPLIT - PH;             // channels of a frame
	constexpr unsigned FB = CHT * 2;                      // bytes per input frame (all channels)
	constexpr unsigned FBL = CH * 2;                      // bytes of one lane's share of a frame
	constexpr unsigned TILE_BYTES = NV * 16u * NTHREADS;
	// PADDED tiles (cr_device.hpp padded_frames: 9-11 and 13-15 channels on the run-time-slot instances, where the plan asks): the DMA always lands in the first
	// buffer, an LDS -> LDS pass repacks it into the second - lane-share L at L * 16 - and the frames read that
	// (PADT: chosen per plan - crhip_poly_launch.padded - where the shorter tiles it leaves are worth it)
	constexpr bool PADL = PADT != 0;
	static_assert(!PADL || (padded_frames<CH, TT, SPLIT, PH>() && !DUAL && ABL == 0), "padded tiles: the run-time-slot instances of 9-11 and 13-15 channels");

	constexpr int ABL_LDS = ABL >> 4;   // (bits 4, 5: the LDS forms)
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const unsigned tid = threadIdx.x;
	unsigned stamp_cycles = 0;
	if constexpr (ABL == 6)
	{
		// diagnostic build only: in-kernel clock = cycles / (ticks / 100 MHz)  (MI355X_MICROARCH.md, DVFS give-back item 6).
		// 32 bits of the cycle counter, and the start tick goes out at once: the instance has to stay at or below 96 SGPRs
		// to be the same kernel as the one it stands for (see the note at `phase` below).
		stamp_cycles = (unsigned)__builtin_amdgcn_s_memtime();
		if (tid == 0 && a.debug_stamps != nullptr)
			a.debug_stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
	}
	const unsigned rows_bytes = a.plane_rows * a.row_stride * 4u;   // planes of plane_rows x 16 bytes
	const int *rows = reinterpret_cast<const int *>(smem);
	unsigned char *tiles = smem + rows_bytes;

	// Tiles of tile_frames output frames are handed out in order: the first gridDim.x tiles by workgroup number, every
	// further one by an atomic ticket (a.d_tickets[0]).  (1) At any moment the resident workgroups stream ONE compact
	// window of the input and of the output, like a flat grid would, instead of gridDim.x far-apart streams: worth ~10 %
	// of HBM throughput for this read:write mix (tools/microbench/streambench.hip).  (2) Workgroups do not run at the
	// same speed - with equal shares the median workgroup finished at 49 us of a 64 us kernel - so whoever is free takes
	// the next tile.  The ticket of the tile AFTER the next one is drawn while the current tile is computed and handed to
	// the other waves through an LDS mailbox, so neither the atomic's latency nor the DMA of the next tile is exposed.
	// The last workgroup to finish zeroes the two counters again: the slot is clean for the next launch (also for a
	// hipGraph replay of this one).
	const uint64_t NT64 = a.tile_frames;
	const uint64_t n_tiles = (a.n_out + NT64 - 1) / NT64;
	if (blockIdx.x >= n_tiles)
		return;
	// (an LDS pointer by TYPE: through a generic pointer the mailbox was written and read with flat instructions, which count in
	// vmcnt as well - and a volatile flat access is followed by vmcnt(0): every wave drained its tile's stores behind the
	// barrier of every tile before it read the next tile's number)
	typedef __attribute__((address_space(3))) volatile unsigned lds_word;
	lds_word *mailbox = (lds_word *)(smem + rows_bytes + 2u * TILE_BYTES);

	const uint64_t in_base = reinterpret_cast<uint64_t>(a.d_in);
	const uint64_t in_end = in_base + a.in_valid_bytes;
	const unsigned T = (TT > 0 ? (unsigned)TT : a.slots) + a.window_extra;   // frames of a tap window, the largest shift included

	// Starts the LDS-DMA of the input window of the tile that begins at output frame jt: 16-byte buffer loads that
	// land directly in `tile` (no register staging, no ds_write), each wave filling a contiguous 1 KiB piece per
	// instruction.  Returns the byte offset of the window's first frame inside the (16-byte aligned) tile image.
	// The loads are NOT waited for here.
	const unsigned wave_first = __builtin_amdgcn_readfirstlane(tid & ~63u);
	auto fetch = [&](uint64_t jt, unsigned n, unsigned char *tile) -> unsigned {
		const uint64_t pos = a.pos0 + jt * (uint64_t)a.increment;
		if constexpr (DUAL)
		{
			// two MONO windows: 16-byte vector c of the tile buffer belongs to the first window while c * 16 < TILE_BYTES / 2, to the
			// second (dual_in_bytes further on in the input) from there on - wave-uniform either way.  Returns both shifts.
			constexpr unsigned HALF = TILE_BYTES / 2u;
			const unsigned last_rel = (unsigned)(((pos & 0xFFFFu) + (uint64_t)(n - 1) * a.increment) >> 16);
			unsigned shifts = 0;
#pragma unroll
			for (int h = 0; h < 2; ++h)
			{
				const uint64_t first_byte = in_base + ((pos >> 16) + a.first_slot) * 2u + (h ? a.dual_in_bytes : 0ull);
				const uint64_t aligned = first_byte & ~(uint64_t)3;   // (any alignment will do for the DMA: the window starts at the word of its first sample)
				const unsigned shift = (unsigned)(first_byte - aligned);
				uint64_t want = (uint64_t)shift + (uint64_t)(last_rel + T) * 2u;
				const uint64_t avail = in_end > aligned ? in_end - aligned : 0;
				if (want > avail)
					want = avail;
				want = (want + 3u) & ~(uint64_t)3u;
				if (want > HALF)
					want = HALF;   // (the host sizes dual launches so that a window fits its half; never reached)
				shifts |= shift << (8 * h);
				const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)aligned);
				const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(aligned >> 32));
				const unsigned rec = __builtin_amdgcn_readfirstlane((unsigned)want);
				const __amdgpu_buffer_rsrc_t rsrc =
				    __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0, (int)rec, 0x00020000);
#pragma unroll
				for (int v = 0; v < NV; ++v)
				{
					// which half this wave's vector lies in is wave-uniform (NV == 1: by thread number; NV == 2: by v)
					const bool second = NV == 1 ? (wave_first >= NTHREADS / 2u) : (v >= NV / 2);
					if (second == (h == 1))
						__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(tile + (v * NTHREADS + wave_first) * 16u), 16,
						                                         (int)((v * NTHREADS + tid) * 16u - (h ? HALF : 0u)), 0, 0, 0);
				}
			}
			return shifts;
		}
		const uint64_t first_byte = in_base + ((pos >> 16) + a.first_slot) * FB;
		const uint64_t aligned = first_byte & ~(uint64_t)15;
		const unsigned shift = (unsigned)(first_byte - aligned);
		// bytes of the window: frames [0, last_rel + T) where last_rel is the last frame's integer advance
		const unsigned last_rel = (unsigned)(((pos & 0xFFFFu) + (uint64_t)(n - 1) * a.increment) >> 16);
		uint64_t want = (uint64_t)shift + (uint64_t)(last_rel + T) * FB;
		uint64_t avail = in_end > aligned ? in_end - aligned : 0;
		if (want > avail)
			want = avail;
		// The descriptor's range check works on whole dwords: a window that ends on a 2-byte boundary (odd channel
		// counts, mono) would lose its last sample.  Rounding up stays inside the same aligned dword, hence inside the
		// same page as the last valid sample; the extra half-dword only ever meets a zero weight.
		want = (want + 3u) & ~(uint64_t)3u;
		// wave-uniform descriptor: base = aligned window start, num_records = bytes we may touch (loads beyond it
		// deliver zeros, which only ever meet zero weights)
		const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)aligned);
		const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(aligned >> 32));
		const unsigned rec = __builtin_amdgcn_readfirstlane((unsigned)want);
		const __amdgpu_buffer_rsrc_t rsrc =
		    __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0, (int)rec, 0x00020000);
#pragma unroll
		for (int v = 0; v < NV; ++v)
			if (!((ABL & 15) == 2 || (ABL & 15) == 3) || a.n_out == 1)
			__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(tile + (v * NTHREADS + wave_first) * 16u), 16,
			                                         (int)((v * NTHREADS + tid) * 16u), 0, 0, 0);
		return shift;
	};

	// vmcnt counts loads, LDS-DMA and stores together, in issue order.  The DMA of the NEXT tile is issued before
	// this tile's stores, so waiting until only this tile's stores are outstanding means the DMA has landed, while
	// the stores stay in flight across the barrier.  The count must be a literal: full tiles of 4, 2 or 1 groups run
	// as straight-line code for that reason; a ragged tile drains everything.
	constexpr unsigned GROUP = NTHREADS * U;
	constexpr int ADJ = 0;   // a lane's U frames are NTHREADS apart: every store instruction is coalesced across the wave
	// (a LOWER bound of the store instructions per group: see min_stores_of_bytes; a phantom lane-frame leaves as at least one)
	constexpr int STORES_PER_GROUP = PH ? U : (DUAL ? 2 * U : U * min_stores_of_bytes(CH * (OUT16 ? 2 : 4)));

	// Tickets.  One global counter would serialise: a single word sustains ~88 atomic draws per microsecond on this
	// chip (MI355X_MICROARCH.md, "dequeue") and a 10-minute stereo launch draws 7,000 of them - measured 92 us instead
	// of 64.  So there are LANES counters (each on its own 128-byte line); the tiles are dealt round-robin to LANES
	// sequences, workgroup b belongs to sequence b % LANES (workgroups b and b + 8 are observed to share an XCD, so with
	// 8 lanes a sequence is mostly one XCD's - a speed matter only), starts with the tile of its own number and then
	// draws from its sequence's counter.  Every sequence has at least one workgroup (LANES <= gridDim.x), so every tile
	// is computed wherever the workgroups land.  Only thread 0 of the workgroup draws.
	const unsigned LANES = gridDim.x < 8u ? gridDim.x : 8u;
	const unsigned lane_id = blockIdx.x % LANES;
	const uint64_t lane_tiles = (n_tiles - lane_id + LANES - 1u) / LANES;            // tiles of this sequence
	const unsigned lane_groups = (gridDim.x - lane_id + LANES - 1u) / LANES;          // its workgroups = its pre-assigned tiles
	unsigned *lane_counter = a.d_tickets + lane_id * 32u;
	// (Helping other sequences out once the own one is exhausted was tried: deciding where to draw needs the counter's
	// value NOW, and a dependent load at the top of every tile stalls wave 0 - and with it the workgroup - for a memory
	// round trip per tile: 148 us instead of 64.  The draw below has no consumer until the end of the tile.)
	// The draw is a scalar atomic (draw_ticket, cr_device.hpp), made by wave 0 alone (wave-uniform branch).  As a vector atomic
	// of thread 0 it cost wave 0 a vmcnt(0) per tile - every store of the previous tile and the DMA just issued - which was the
	// "issuing the next tile's DMA" share of a tile in the stamped diagnostic instance (42-44 % of wave 0's cycles, the other
	// waves waiting for it at the barrier).
	auto draw = [&]() -> unsigned {
		const uint64_t k = (uint64_t)lane_groups + draw_ticket(lane_counter);
		return k < lane_tiles ? (unsigned)(lane_id + LANES * k) : 0xFFFFFFFFu;
	};
	const bool wave0 = __builtin_amdgcn_readfirstlane((int)(tid >> 6)) == 0;
	// a workgroup that has drawn a ticket beyond its sequence is done drawing; the last such workgroup zeroes the slot
	auto retire = [&]() {
		if (tid == 0)
		{
			unsigned *finished = a.d_tickets + 8u * 32u;
			if (__hip_atomic_fetch_add(finished, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u)
			{
				for (unsigned c = 0; c < 8u; ++c)
					__hip_atomic_store(a.d_tickets + c * 32u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(finished, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
	};

	uint64_t tile_index = blockIdx.x;
	uint64_t jt = tile_index * NT64;
	unsigned n = (unsigned)((a.n_out - jt < NT64) ? (a.n_out - jt) : NT64);
	unsigned shift = fetch(jt, n, tiles);
	// a.dynamic_tiles == 0: plain round-robin (tile + gridDim.x), no tickets - for configurations whose tiles are so
	// small that a ticket and a mailbox hand-over per tile cost more than the imbalance they remove (8-channel frames)
	const bool dynamic = a.dynamic_tiles != 0;
	if (dynamic && wave0)
	{
		const unsigned first_ticket = draw();
		if (tid == 0)
			mailbox[0] = first_ticket;
	}
	// stage the polyphase rows once per workgroup (L2-resident after the first workgroups) - AFTER the first tile's DMA and
	// the first ticket are on their way, so that the three round trips of a workgroup's start overlap
	{
		const unsigned nvec = rows_bytes / 16u;
		const u32x4 *src = reinterpret_cast<const u32x4 *>(a.d_rows);
		u32x4 *dst = reinterpret_cast<u32x4 *>(smem);
		// SWZ: the rows land rotated within their block of 16 (one_frame / fetch_frame read them back the same way): the image in
		// global memory is the plain one, shared by plans of every increment; the rotation that suits THIS increment is the
		// plan's (cr_poly_pick_swizzle)
		auto place = [&](unsigned r) { return SWZ ? ((r & ~15u) | ((__umul24(r >> 4, a.swizzle) + r) & 15u)) : r; };
		if constexpr (SWZ && (ASM & 0xFF) != 2)
		{
			const unsigned planes = a.row_stride / 4u;
			for (unsigned q = 0; q < planes; ++q)
				for (unsigned r = tid; r < a.plane_rows; r += NTHREADS)
					dst[q * a.plane_rows + place(r)] = src[q * a.plane_rows + r];
		}
		else if constexpr ((ASM & 0xFF) == 2)
		{
			// the 64-bit chain takes its weights as magnitudes, << 15 outside the two centre slots (mad_staged_weight, cr_device.hpp)
			// (every load first, then the stores: a row image of 1,025 rows is two rows per thread and plane, and one round trip
			// instead of four is 0.4 us of every launch)
			constexpr int PL = (TT + 1 + 3) / 4;
			auto staged = [&](u32x4 v, int q) {
				v.x = (unsigned)mad_staged_weight<TT, ((unsigned)ASM >> 8)>((int)v.x, 4 * q);
				v.y = (unsigned)mad_staged_weight<TT, ((unsigned)ASM >> 8)>((int)v.y, 4 * q + 1);
				v.z = (unsigned)mad_staged_weight<TT, ((unsigned)ASM >> 8)>((int)v.z, 4 * q + 2);
				v.w = (unsigned)mad_staged_weight<TT, ((unsigned)ASM >> 8)>((int)v.w, 4 * q + 3);
				return v;
			};
			if (a.plane_rows <= 2u * NTHREADS && a.row_stride == 4u * PL)
			{
				u32x4 v[PL][2];
#pragma unroll
				for (int q = 0; q < PL; ++q)
#pragma unroll
					for (int k = 0; k < 2; ++k)
					{
						const unsigned r = tid + (unsigned)k * NTHREADS;
						if (r < a.plane_rows)
							v[q][k] = src[(unsigned)q * a.plane_rows + r];
					}
#pragma unroll
				for (int q = 0; q < PL; ++q)
#pragma unroll
					for (int k = 0; k < 2; ++k)
					{
						const unsigned r = tid + (unsigned)k * NTHREADS;
						if (r < a.plane_rows)
							dst[(unsigned)q * a.plane_rows + place(r)] = staged(v[q][k], q);
					}
			}
			else
			{
				const unsigned planes = a.row_stride / 4u;
				for (unsigned q = 0; q < planes; ++q)
					for (unsigned r = tid; r < a.plane_rows; r += NTHREADS)
						dst[q * a.plane_rows + place(r)] = staged(src[q * a.plane_rows + r], (int)q);
			}
		}
		else
		{
			for (unsigned i = tid; i < nvec; i += NTHREADS)
				dst[i] = src[i];
		}
	}
	asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
	__syncthreads();   // rows staged (plain stores to LDS), first tile landed and first ticket posted, for every wave
	uint64_t next_index = dynamic ? __builtin_amdgcn_readfirstlane(mailbox[0]) : tile_index + gridDim.x;   // wave-uniform

	// diagnostic instance (ABL == 6) only: where a tile's cycles go, summed over the tiles of this workgroup as seen by
	// wave 0 - [0] issuing the next tile's DMA + ticket, [1] arithmetic + stores, [2] waiting for the DMA (vmcnt),
	// [3] waiting at the barrier (+ mailbox)
	// The sums live in VGPRs on purpose: as wave-uniform 64-bit values they took the instance from 80 to 106 SGPRs, and above
	// 96 SGPRs a SIMD holds 7 waves instead of 8 - one 1024-thread workgroup per CU instead of two, i.e. a different kernel.
	unsigned phase[4] = {0, 0, 0, 0};
	unsigned t_mark = 0;
	auto mark = [&](int which) {
		if constexpr (ABL == 6)
		{
			__builtin_amdgcn_sched_barrier(0);
			const unsigned now = (unsigned)__builtin_amdgcn_s_memtime();
			__builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): s_memtime returns through the scalar data path
			if (which >= 0)
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(phase[which]) : "s"(now - t_mark));
			t_mark = now;
			__builtin_amdgcn_sched_barrier(0);
		}
	};
	mark(-1);

	for (unsigned it = 0;; ++it)
	{
		// (dual mono: the DMA always lands in the first buffer, the second holds the interleaved tile the frames read)
		// (its distance as a VALUE, not a constant: hipcc otherwise adds it to the address of every window read that has no room for
		// it in its offset field - two VALU per frame)
		unsigned dual_at = TILE_BYTES;
		asm volatile("" : "+s"(dual_at));
		const unsigned char *tile = (DUAL || PADL) ? tiles + dual_at : tiles + (it & 1u) * TILE_BYTES;
		const bool more = next_index < n_tiles;
		const uint64_t jn = next_index * NT64;
		unsigned n_next = 0, shift_next = 0;
		unsigned ticket = 0;

		if constexpr (DUAL)
		{
			// The tile's two mono windows have landed in the two halves of the first buffer (the barrier behind us): interleave them
			// into a stereo tile, frame f = {first window's frame f, second window's frame f}.  A thread takes PAIRS of frames: the
			// aligned dwords around each window's pair, a funnel shift where the window starts in the middle of a dword
			// (wave-uniform), byte permutes, one wide write.  Then a second barrier: the interleaved tile is complete and the
			// DMA buffer free for the next tile's windows.
			const uint64_t pos_t = a.pos0 + jt * (uint64_t)a.increment;
			const unsigned frames = (unsigned)(((pos_t & 0xFFFFu) + (uint64_t)(n - 1) * a.increment) >> 16) + T;
			// (the dual fetches start at the dword of a window's first sample: the shifts are 0 or 2, both halves 16-byte aligned)
			const unsigned ra = (shift & 2u) * 8u, rb = ((shift >> 8) & 2u) * 8u;
			const unsigned *wa = reinterpret_cast<const unsigned *>(tiles);
			const unsigned *wb = reinterpret_cast<const unsigned *>(tiles + TILE_BYTES / 2u);
			i32x4 *inter = reinterpret_cast<i32x4 *>(tiles + TILE_BYTES);
			// FOUR frames per thread (one pass for a 16 KiB tile: at most 1024 quads): 8 + 4 bytes of each window, an 16-byte write
			for (unsigned q = tid; q < (frames + 3u) / 4u; q += NTHREADS)
			{
				const i32x2 a01 = *reinterpret_cast<const i32x2 *>(wa + 2u * q), b01 = *reinterpret_cast<const i32x2 *>(wb + 2u * q);
				const unsigned a2 = wa[2u * q + 2u], b2 = wb[2u * q + 2u];
				const unsigned ap0 = __builtin_amdgcn_alignbit((unsigned)a01.y, (unsigned)a01.x, ra), ap1 = __builtin_amdgcn_alignbit(a2, (unsigned)a01.y, ra);
				const unsigned bp0 = __builtin_amdgcn_alignbit((unsigned)b01.y, (unsigned)b01.x, rb), bp1 = __builtin_amdgcn_alignbit(b2, (unsigned)b01.y, rb);
				i32x4 f;
				f.x = (int)__builtin_amdgcn_perm(bp0, ap0, 0x05040100u);   // {a.lo, b.lo}
				f.y = (int)__builtin_amdgcn_perm(bp0, ap0, 0x07060302u);   // {a.hi, b.hi}
				f.z = (int)__builtin_amdgcn_perm(bp1, ap1, 0x05040100u);
				f.w = (int)__builtin_amdgcn_perm(bp1, ap1, 0x07060302u);
				inter[q] = f;
			}
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
		}

		if constexpr (PADL)
		{
			// The tile's window has landed in the first buffer (the barrier behind us).  Lane-share L = 2 * frame + half: CH * 2 bytes
			// from byte shift + frame * FB + half * FBL - any 2-byte boundary - as the aligned dwords that cover it, funnel-shifted, one
			// 16-byte write to L * 16 of the second buffer.  Then a second barrier: the padded tile is complete and the first buffer
			// free for the next tile's DMA.  (The host sizes these tiles for 32 bytes per frame: plan_geometry.)
			const uint64_t pos_t = a.pos0 + jt * (uint64_t)a.increment;
			const unsigned shares = 2u * ((unsigned)(((pos_t & 0xFFFFu) + (uint64_t)(n - 1) * a.increment) >> 16) + T);
			constexpr int WORDS = (CH + 1) / 2;
			i32x4 *padded = reinterpret_cast<i32x4 *>(tiles + TILE_BYTES);
			for (unsigned L = tid; L < shares; L += NTHREADS)
			{
				const unsigned at = shift + (L >> 1) * FB + (L & 1u) * FBL;
				const unsigned *q = reinterpret_cast<const unsigned *>(tiles + (at & ~3u));
				const unsigned r = (at & 2u) * 8u;
				unsigned d[WORDS + 1];
#pragma unroll
				for (int k = 0; k < WORDS + 1; ++k)
					d[k] = q[k];
				int v[4] = {0, 0, 0, 0};
#pragma unroll
				for (int k = 0; k < WORDS; ++k)
					v[k] = (int)__builtin_amdgcn_alignbit(d[k + 1], d[k], r);
				i32x4 e;
				e.x = v[0];
				e.y = v[1];
				e.z = v[2];
				e.w = v[3];
				padded[L] = e;
			}
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
		}

		if (more)
		{
			// the other buffer was last read in the previous iteration, which every wave has left (barrier below)
			n_next = (unsigned)((a.n_out - jn < NT64) ? (a.n_out - jn) : NT64);
			shift_next = fetch(jn, n_next, (DUAL || PADL) ? tiles : tiles + ((it + 1u) & 1u) * TILE_BYTES);
			if (dynamic && wave0)
				ticket = draw();   // for the tile after the next one; posted below, just before the barrier
		}
		mark(0);

		const uint64_t pos = a.pos0 + jt * (uint64_t)a.increment;
		const unsigned frac0 = (unsigned)(pos & 0xFFFFu);
		// lane-frames: a frame shared by SPLIT lanes counts SPLIT times; lane-frame L is lane share L % SPLIT of frame L / SPLIT
		const unsigned nl = n * SPLIT;
		int *out_tile = reinterpret_cast<int *>(a.d_out) + jt * CHT;             // OUT16 == 0
		short *out_tile16 = reinterpret_cast<short *>(a.d_out) + jt * CHT;       // OUT16 == 1
		// phantom instances: lane-frame L is share L % 2 of frame L / 2; the shares are CH and CH - 1 channels, stored sample by
		// sample (an odd channel count leaves nothing wider aligned), the last one only by the first lane of a pair - one store
		// instruction per wave either way, so the counted vmcnt below holds
		auto store_phantom = [&](unsigned L, const int *v) {
			const size_t at = (size_t)(L >> 1) * CHT + (L & 1u) * CH;
			if constexpr (OUT16)
			{
#pragma unroll
				for (int c = 0; c < CH; ++c)
				{
					if (c == CH - 1 && (L & 1u))
						break;
					out_tile16[at + c] = (short)clamp_s16(v[c]);
				}
			}
			else
			{
				// (plain stores whatever NT says: the two shares of a frame leave in pieces that do not tile 16-byte granules -
				// see nt_suits_ints, cr_device.hpp)
				store_ints_dword_aligned<CH - 1, 0>(out_tile + at, v);
				if (!(L & 1u))
					out_tile[at + CH - 1] = v[CH - 1];
			}
		};
		const unsigned char *base = DUAL ? tile : (PADL ? tile + (tid % SPLIT) * 16u : tile + shift + (tid % SPLIT) * FBL);
		// dual mono: where this tile's first frames go, and how many of its SECOND frames exist (the second half of the stream may be
		// the shorter one)
		int *out_mono = reinterpret_cast<int *>(a.d_out) + jt;
		const unsigned jt32 = (unsigned)jt;   // (dual mono: fewer than 2^30 pairs)
		const unsigned dual_valid = !DUAL ? 0u : (a.dual_valid_frames > jt32 ? ((a.dual_valid_frames - jt32 < (unsigned)NT64) ? (a.dual_valid_frames - jt32) : (unsigned)NT64) : 0u);
		// Stores through ONE buffer descriptor from the tile's first frames to the end of its second frames, with a wave-uniform frame
		// offset (scalar: + H frames for the second) and one constant lane offset: no address arithmetic and no predicate per frame -
		// a second frame that does not exist lies beyond the descriptor's range and the store is dropped.  (H < 2^30 frames: the host.)
		const uint64_t lo_base = reinterpret_cast<uint64_t>(out_mono);
		const unsigned second_bytes = a.dual_out_frames * 4u;   // wave-uniform
		const unsigned valid_here = dual_valid < n ? dual_valid : n;
		const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
		    reinterpret_cast<void *>(((uint64_t)__builtin_amdgcn_readfirstlane((int)(unsigned)(lo_base >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)lo_base)), 0,
		    (int)__builtin_amdgcn_readfirstlane((int)(!DUAL ? 0u : (valid_here != 0u ? second_bytes + valid_here * 4u : n * 4u))), 0x00020000);
		const unsigned lane_bytes = tid * 4u;
		auto store_dual = [&](unsigned first, const int *v) {   // first: wave-uniform (the frame of lane 0 of the workgroup)
			unsigned at = first * 4u;
			asm volatile("" : "+s"(at));   // (the instruction's scalar offset: as a compile-time constant hipcc adds it to the lane offset, one VALU per store)
			__builtin_amdgcn_raw_buffer_store_b32(v[0], rs_out, (int)lane_bytes, (int)at, NT ? 2 : 0);
			__builtin_amdgcn_raw_buffer_store_b32(v[1], rs_out, (int)lane_bytes, (int)(at + second_bytes), NT ? 2 : 0);
		};
		auto store_dual_lane = [&](unsigned frame, const int *v) {   // frame: per lane (ragged tiles)
			out_mono[frame] = v[0];
			if (frame < dual_valid)
				out_mono[a.dual_out_frames + frame] = v[1];
		};
		// One group = NTHREADS * U frames: U independent frames per lane, no bounds checks.
		// Positions are formed as (lane part, once per tile) + (group part, wave-uniform, scalar unit): one VALU add per
		// frame instead of a 24-bit multiply-add; same for the output address, which goes out as SGPR base + lane offset.
		const unsigned lane_rel = __umul24(tid / SPLIT, a.increment) + frac0;
		auto group = [&](unsigned g) {
			int outv[U * CH];
#pragma unroll
			for (int u = 0; u < U; ++u)
			{
				const unsigned first = g + u * NTHREADS;   // wave-uniform: frame of lane 0 of the workgroup
				if constexpr (ABL == 4)
				{
#pragma unroll
					for (int c = 0; c < CH; ++c)
						outv[u * CH + c] = (int)(first + tid);
				}
				else
					one_frame<CH, TT, MODE, NORM, ASM, SWZ, SPLIT, PH, PADT>(a, rows, base, lane_rel + (first / SPLIT) * a.increment, outv + u * CH);
			}
			if constexpr ((ABL & 15) == 1 || (ABL & 15) == 3)
			{
				// keep the arithmetic alive without the stores (cdna_hip_programming.md rule 17)
#pragma unroll
				for (int c = 0; c < U * CH; ++c)
					asm volatile("" ::"v"(outv[c]));
				return;
			}
#pragma unroll
			for (int u = 0; u < U; ++u)
			{
				if constexpr (PH)
				{
					store_phantom(g + u * NTHREADS + tid, outv + u * CH);
				}
				else if constexpr (DUAL)
				{
					store_dual(g + u * NTHREADS, outv + u * CH);
				}
				else if constexpr (OUT16)
				{
					short *group_out = out_tile16 + (size_t)(g + u * NTHREADS) * CH;   // uniform
					store_shorts<CH, NT>(group_out + tid * CH, outv + u * CH);
				}
				else
				{
					int *group_out = out_tile + (size_t)(g + u * NTHREADS) * CH;   // uniform
					store_ints<CH, NT>(group_out + tid * CH, outv + u * CH);
				}
			}
		};

		// Full tiles of 4, 3, 2 or 1 groups as straight-line code (see the vmcnt note above).  Specialised instances
		// (TT > 0) run the frames of a tile as a software pipeline: the LDS reads of frame i+1 are issued before the
		// arithmetic of frame i.
		auto run_groups = [&](auto groups_tag) {
			constexpr int G = decltype(groups_tag)::value;
			constexpr int N = G * U;   // frames per lane in this tile

			if constexpr (TT > 0 && (ABL == 0 || ABL == 6 || ABL >= 16))
			{
				FrameData<CH, TT> d[2];
				fetch_frame<CH, TT, MODE, SWZ, SPLIT, PH, ABL_LDS>(a, rows, base, lane_rel, d[0]);
#pragma unroll
				for (int i = 0; i < N; ++i)
				{
					const unsigned first = (unsigned)(i / U) * GROUP + (unsigned)(i % U) * NTHREADS;   // uniform
					int outv[CH];

					if (i + 1 < N)
					{
						const unsigned next_first = (unsigned)((i + 1) / U) * GROUP + (unsigned)((i + 1) % U) * NTHREADS;
						fetch_frame<CH, TT, MODE, SWZ, SPLIT, PH, ABL_LDS>(a, rows, base, lane_rel + (next_first / SPLIT) * a.increment, d[(i + 1) & 1]);
					}
					__builtin_amdgcn_sched_barrier(0);   // keep the reads above the arithmetic below
					compute_frame<CH, TT, NORM, ASM>(d[i & 1], outv);

					if constexpr ((ABL & 15) == 1 || (ABL & 15) == 3)
					{
#pragma unroll
						for (int c = 0; c < CH; ++c)
							asm volatile("" ::"v"(outv[c]));
					}
					else if constexpr (DUAL)
						store_dual(first, outv);
					else if constexpr (PH)
						store_phantom(first + tid, outv);
					else if constexpr (OUT16)
						store_shorts<CH, NT>(out_tile16 + (size_t)first * CH + tid * CH, outv);
					else
						store_ints<CH, NT>(out_tile + (size_t)first * CH + tid * CH, outv);
					__builtin_amdgcn_sched_barrier(0);
				}
			}
			else
			{
#pragma unroll
				for (int gi = 0; gi < G; ++gi)
					group((unsigned)gi * GROUP);
			}

			mark(1);
			if constexpr (G * STORES_PER_GROUP <= 63)
				asm volatile("s_waitcnt vmcnt(%0)" ::"i"(G * STORES_PER_GROUP) : "memory");
			else
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			mark(2);
		};

		if (nl == 4u * GROUP)
			run_groups(std::integral_constant<int, 4>());
		else if (nl == 3u * GROUP)
			run_groups(std::integral_constant<int, 3>());
		else if (nl == 2u * GROUP)
			run_groups(std::integral_constant<int, 2>());
		else if (nl == GROUP)
			run_groups(std::integral_constant<int, 1>());
		else
		{
			// ragged tile (only the stream's last tile can be one)
			const unsigned n_full = nl - nl % GROUP;
			unsigned g = 0;
			for (; g < n_full; g += GROUP)
				group(g);
			for (unsigned jl = g + tid; jl < nl; jl += NTHREADS)
			{
				int outv[CH];
				one_frame<CH, TT, MODE, NORM, ASM, SWZ, SPLIT, PH, PADT>(a, rows, base, __umul24(jl / SPLIT, a.increment) + frac0, outv);
				if constexpr (PH)
					store_phantom(jl, outv);
				else if constexpr (DUAL)
					store_dual_lane(jl, outv);
				else if constexpr (OUT16)
					store_shorts<CH, NT>(out_tile16 + (size_t)jl * CH, outv);
				else
					store_ints<CH, NT>(out_tile + (size_t)jl * CH, outv);
			}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}

		if constexpr (ABL == 6)
		{
			// per tile of every workgroup (first 32): tile index << 48 | tick at which wave 0 had issued the tile's last store
			if (tid == 0 && a.debug_stamps != nullptr && it < 32u)
				a.debug_stamps[4 * 4096 + 4 * 64 + 32 * blockIdx.x + it] = (tile_index << 48) | (__builtin_amdgcn_s_memrealtime() & 0xFFFFFFFFFFFFull);
		}
		if (!more)
		{
			if constexpr (ABL == 6)
			{
				if (tid == 0 && a.debug_stamps != nullptr)
				{
					// per workgroup: {shader cycles of its lifetime, start tick, end tick, XCC id}
					a.debug_stamps[4 * blockIdx.x + 0] = (unsigned)__builtin_amdgcn_s_memtime() - stamp_cycles;
					a.debug_stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
					a.debug_stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg(63508 /* HW_REG_XCC_ID, bits 0..3 */) & 0xF;
					if (blockIdx.x < 64u)
						for (int k = 0; k < 4; ++k)
							a.debug_stamps[4 * 4096 + 4 * blockIdx.x + k] = phase[k];
				}
			}
			if (dynamic)
				retire();
			break;
		}

		// every wave's share of the next tile has landed once every wave is past its wait; the mailbox has two slots,
		// used alternately, so that a slot is never rewritten before every wave has read it
		if (dynamic && tid == 0)
			mailbox[(it + 1u) & 1u] = ticket;
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		mark(3);
		tile_index = next_index;
		jt = jn;
		n = n_next;
		shift = shift_next;
		next_index = dynamic ? __builtin_amdgcn_readfirstlane(mailbox[(it + 1u) & 1u]) : tile_index + gridDim.x;
	}
}

} // namespace

#endif // CR_KPOLY_HPP
