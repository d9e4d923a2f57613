// cr_kwave2.hpp - k_wave2: k_wave's streaming with the tap arithmetic of k_up2, 1 to 8 channels, compile-time or run-time slot count.
#ifndef CR_KWAVE2_HPP
#define CR_KWAVE2_HPP

#include "cr_device.hpp"

namespace
{

// ---------------------------------------------------------------------------------------------------------
// k_wave2 - one lane per output frame (as k_poly / k_wave), 2 VALU per tap and channel instead of 4
// ---------------------------------------------------------------------------------------------------------
// k_poly and k_wave keep the input window packed (two int16 per dword) and spend four SDWA instructions per tap and channel:
// multiply, sign, +0xFFFF where negative, add the high word.  Here the wave first EXPANDS its window once per wave-tile into a
// second LDS buffer of one dword per sample (a shift or a mask per sample, amortised over the output frames that read it),
// and the tap is the 64-bit multiply-add of k_up2 on an accumulator pair {lo, hi}:
//     lo = arm                           anything in [2^32 - 65536, 2^32) where the product can be negative, in [0, 65536) where not
//     {lo, hi} = X * W + {lo, hi}        X * W = sample * weight << 16: integer part into hi, fraction bits in the top of lo;
//                                        such a lo carries exactly when a negative product has a fraction: C's truncation
//                                        (clownresampler.h:1020 via :625)
// Three ways to arm, by what is known about the rows:
//   SIGNED == 1 (any rows): X = sample << 16, W = weight as it is, lo = sext(top byte of X) ^ sext(top byte of W) in ONE SDWA
//                instruction - bits 31..7 the product's sign, the low 7 bits noise the carry does not see.  (The first form,
//                (X ^ W) >> 31 in two instructions, made the tap 3.)  Also the run-time-slot form (TT == 0, below).
//   SIGNED == 0 (pure upsampling: the sign of a slot's weights is a compile-time property, NEGMASK, checked by the host per
//                plan as for k_up): rows staged as magnitudes, the slots with negative weights accumulate into a second pair,
//                the two sums subtracted at the end (truncation toward zero is odd-symmetric); lo = X >> 31 - or, with
//   SAFEMASK != 0, the mov-armed form (below): X = 2 * sample is its own lo.
// The streaming - a private double-buffered LDS-DMA window per wave, counted vmcnt, coalesced non-temporal stores, tickets over
// 32 counter lanes (for the tail of a launch only) - is k_wave's.  LDS per wave: two packed windows (NVW KiB each) and one
// expanded window (2 x NVW KiB).
// (Loading the packed window into registers instead of LDS - coalesced buffer loads a wave-tile ahead, the expansion straight
// from registers, half the LDS footprint - was tried and is not kept: with hipcc's own loads it protects the registers with
// vmcnt(0), draining the tile's stores every time (117 against 105 us on the 8-lobe 44.1 -> 48 kHz workload), and inline-assembly
// loads cannot be made safe: hipcc copies their destination registers before the data has landed.)
//   WAVES  waves per workgroup     NVW  1 KiB DMA pieces per wave-tile     ITER  frames per lane per wave-tile
// ---------------------------------------------------------------------------------------------------------
// LDS reads a frame has in flight BEHIND those of plane q while `ahead` planes are requested in advance: the planes q + 1 ...
// q + ahead - 1 (a plane = the row vector, for the first channel pair, + one read per slot and channel read)
constexpr int wave2_reads_ahead(bool with_rows, int reads_per_slot, int tt, int q, int ahead, int planes)
{
	int later = 0;
	for (int g = q + 1; g < planes && g < q + ahead; ++g)
		later += (with_rows ? 1 : 0) + reads_per_slot * ((tt - 4 * g) < 4 ? (tt - 4 * g) : 4);
	return later;
}

// DUAL = 1: DUAL MONO (see k_poly): a stereo instance run on a MONO stream - output frames j and j + H, whose fractions are equal,
// as its two channels.
// ABL: 0 in every shipped instance.  Timing-only forms (WRONG results; built with -DCRA_WITH_W2_FORMS, reached through
// CLOWNRESAMPLER_AMD_W2_FORM): 1 = every lane reads the window of the wave-tile's first frame (window reads without bank conflicts),
// 2 = every lane reads row 0 (row reads without), 3 = both.
// (Tried in round 5 and NOT kept: the frames of a full wave-tile as ONE pipeline of LDS reads - the first planes of frame i + 1 requested
// while the last planes of frame i are multiplied, into the registers those planes were consumed from; bit-exact, no extra registers -
// within +-1.5 % of frame by frame on every shape, profiles/r05_kwave2_frame_pipeline_ab.log: the waves of a SIMD already cover each
// other's round trips; what binds is the SUM of LDS and VALU cycles, profiles/r05_kwave2_lds_forms.log.)  The wave-tile's two mono windows are fetched into the two halves of the packed buffer by ONE descriptor
// (the second window's lanes add its distance to their offsets), the expansion pass writes them interleaved - X of the first
// window's sample p into dword 0 of entry p + 1, of the second window's into dword 1 - and the frames leave as two 4-byte stores
// through one descriptor that ends where the second half of the stream does.
template <int CH, int TT, int MODE, int NORM, int WAVES, int NVW, int ITER, int OUT16, int NT, unsigned NEGMASK, int SIGNED, unsigned SAFEMASK = 0, int DUAL = 0, int ABL = 0>
__global__ __launch_bounds__(WAVES * 64) void k_wave2(const crhip_poly_launch a)
{
	static_assert(CH >= 1 && CH <= 8, "one lane per frame");
	static_assert(!DUAL || (CH == 2 && TT > 0 && OUT16 == 0), "dual mono: a stereo instance with a compile-time slot count, int32 output");
	// SAFEMASK != 0 (fixed signs only): the MOV-ARMED form.  Of all the one-instruction ways to arm `lo`, a plain v_mov_b32 is the
	// only one that costs next to nothing beside the multiply-add (tools/microbench/chainbench.hip: 231 cycles per wave-frame for
	// the multiply-adds alone, 258 with a v_mov_b32 each, 313-348 with v_and / v_ashrrev / v_not / v_mov_sdwa / v_xor_sdwa / v_bfe).
	// So the window is expanded to X = 2 * sample (sign-extended) instead of sample << 16 and the rows are staged as
	// |weight| << 15: X * W is the same sample * |weight| * 65536, and X ITSELF is a valid `lo` - in [2^32 - 65536, 2^32) where
	// the sample is negative, in [0, 65536) where it is not, which is all the carry needs (the product is a multiple of 65536).
	// |weight| << 15 needs |weight| < 65536; the slots in SAFEMASK - the two around the kernel's centre, whose weights reach
	// 65536 in one row each - keep the plain |weight| and take X << 15 = sample << 16 (one more shift); the host checks the
	// plan's rows against the mask.
	constexpr bool MOVARM = SAFEMASK != 0;
	static_assert(!MOVARM || (!SIGNED && NEGMASK != 0 && TT > 0), "the mov-armed form is for fixed slot signs");
	// TT == 0: the run-time-slot form - slot count, window pieces and waves per workgroup come with the launch (a.slots,
	// a.vecs - 150, blockDim.x / 64), the rows in the run-time-slot image (ceil(slots / 4) zero-padded planes of weights, then
	// one plane with the reciprocal: cr_plan.c), any rows (SIGNED), one frame per lane and wave-tile
	constexpr bool RT = TT == 0;
	static_assert(!RT || (SIGNED == 1 && ITER == 1 && MODE == CRHIP_ROWMODE_AFFINE), "run-time-slot form");
	constexpr unsigned FB = CH * 2;
	const unsigned n_waves = RT ? (blockDim.x >> 6) : (unsigned)WAVES;
	const unsigned NTHREADS = n_waves * 64u;
	const unsigned slots = RT ? a.slots : (unsigned)TT;
	constexpr unsigned WT = 64u * ITER;            // frames per wave-tile
	// wave-tiles per chunk (ticket): 4 for long launches, fewer where a launch would otherwise leave a wave only a handful of
	// chunks (the host halves tile_frames until every wave has ~8: cr_plan_launch); always a power of two
	const unsigned CHUNK = a.tile_frames;
	const unsigned CW = CHUNK / WT;
	const unsigned chunk_shift = (unsigned)__builtin_ctz(CHUNK);
	const unsigned nvw = RT ? a.vecs - 150u : (unsigned)NVW;
	const unsigned BUF = nvw * 1024u;              // bytes per packed window
	const unsigned XBUF = 2u * BUF;                // bytes of the expanded window (4 per sample)
	const unsigned PER_WAVE = 2u * BUF + XBUF;
	constexpr int RS = (TT + 1 + 3) & ~3;          // (specialised form) int32 per row: the slots, the reciprocal, padding
	const unsigned planes_total = RT ? a.row_stride / 4u : (unsigned)(RS / 4);
	constexpr int STORES_PER_FRAME = DUAL ? 2 : min_stores_of_bytes(CH * (OUT16 ? 2 : 4));   // a lower bound: see cr_device.hpp

	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const unsigned tid = threadIdx.x;
	const unsigned lane = tid & 63u;
	const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	int zero119 = 0;   // lives in v119 wherever the taps want it (see one_frame2)
	asm volatile("" : "+v"(zero119));   // (a value, not a constant hipcc would re-materialise in front of every use)

	const unsigned rows_bytes = a.plane_rows * a.row_stride * 4u;
	unsigned char *my_buf = smem + rows_bytes + wave * PER_WAVE;
	unsigned char *my_x = my_buf + 2u * BUF;

	unsigned *waves_done = reinterpret_cast<unsigned *>(smem + rows_bytes + n_waves * PER_WAVE);
	if (tid == 0)
		*waves_done = 0;


	const uint64_t n_chunks = (a.n_out + CHUNK - 1) >> chunk_shift;
	const uint64_t global_wave = (uint64_t)wave * gridDim.x + blockIdx.x;   // (a short launch spreads over the CUs, not over a CU's waves)
	const uint64_t global_waves = (uint64_t)gridDim.x * n_waves;

	// Who takes which chunk.  Chunks go round-robin over the waves (wave w: chunks w, w + waves, ...) for all but the last two
	// rounds - no atomics, nothing to wait for - and the chunks of those last rounds are drawn as tickets, as in k_wave (32 counter
	// lanes; a wave's first chunk of the ticketed region is implicit), so that the waves that got ahead (the XCDs differ by 10 %
	// in clock under load) take more of the tail.  Why not tickets throughout: a returning atomic is a VMEM load into a VGPR,
	// and hipcc waits for such a register at the head of any loop that uses it - vmcnt(0), every store of the previous chunk
	// included - however far apart issue and use are put.  (Worth 1-4 %: profiles/r02_kwave2_trials.log.)
	const uint64_t rounds = n_chunks / global_waves;
	// (a launch of three rounds or fewer is dealt statically altogether: a draw is a memory round trip per chunk, more than the
	// balance of so few chunks is worth)
	const uint64_t static_limit = rounds > 3u ? (rounds - 2u) * global_waves : n_chunks;   // chunks [0, static_limit) are dealt statically
	const uint64_t region_chunks = n_chunks - static_limit;                         // the ticketed region, numbered from 0

	const unsigned LANES = global_waves < 32u ? (unsigned)global_waves : 32u;
	const unsigned lane_id = (unsigned)(global_wave % LANES);
	const uint64_t lane_chunks = region_chunks > lane_id ? (region_chunks - lane_id + LANES - 1u) / LANES : 0;
	const unsigned lane_waves = (unsigned)((global_waves - lane_id + LANES - 1u) / LANES);
	unsigned *lane_counter = a.d_tickets + lane_id * 32u;
	auto draw_issue = [&]() -> unsigned { return draw_ticket(lane_counter); };   // (scalar, the whole wave: cr_device.hpp)
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

	// LDS-DMA of the packed window of the wave-tile of `n` frames starting at output frame `first` into `buf`.  Returns the byte
	// offset of the window's first frame inside the buffer in the low 16 bits and the window's frame count above them.  Not waited for.
	auto fetch = [&](uint64_t first, unsigned n, unsigned char *buf) -> unsigned {
		const uint64_t pos = a.pos0 + first * (uint64_t)a.increment;
		if constexpr (DUAL)
		{
			// returns shift of the first window | shift of the second << 8 | frames << 16
			const uint64_t byte_a = in_base + ((pos >> 16) + a.first_slot) * 2u, byte_b = byte_a + a.dual_in_bytes;
			// (the DMA takes a source of any alignment - tools/microbench/dmaalign.hip - so a window starts at the 4-byte word of its
			// first sample: a shift of 0 or 2 bytes, and a mono window of the longest tile fits half the stereo instance's buffer)
			const uint64_t aligned_a = byte_a & ~(uint64_t)3, aligned_b = byte_b & ~(uint64_t)3;
			const unsigned shift_a = (unsigned)(byte_a - aligned_a), shift_b = (unsigned)(byte_b - aligned_b);
			const unsigned last_rel = (unsigned)(((pos & 0xFFFFu) + (uint64_t)(n - 1) * a.increment) >> 16);
			const unsigned frames = last_rel + slots + a.window_extra;
			// one descriptor from the first window's start to the end of the caller's buffer (whole dwords: see k_poly); what a lane of
			// either window reads beyond it comes back as zeros
			uint64_t avail = in_end > aligned_a ? in_end - aligned_a : 0;
			avail = (avail + 3u) & ~(uint64_t)3u;
			if (avail > 0xFFFFFFFCull)
				avail = 0xFFFFFFFCull;
			const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)aligned_a);
			const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(aligned_a >> 32));
			const unsigned rec = __builtin_amdgcn_readfirstlane((unsigned)avail);
			const unsigned delta = __builtin_amdgcn_readfirstlane((unsigned)(aligned_b - aligned_a));   // (< 2^32: the host)
			const __amdgpu_buffer_rsrc_t rsrc =
			    __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0, (int)rec, 0x00020000);
			const unsigned half = BUF / 2u;
#pragma unroll
			for (int v = 0; v < NVW; ++v)
			{
				// buffer byte q belongs to the first window while q < half, to the second from there on
				const unsigned q = (unsigned)v * 1024u + lane * 16u;
				const unsigned from = q < half ? q : delta + (q - half);
				__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(buf + v * 1024u), 16, (int)from, 0, 0, 0);
			}
			return (unsigned)__builtin_amdgcn_readfirstlane((int)(shift_a | (shift_b << 8) | (frames << 16)));
		}
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
		if constexpr (RT)
		{
			for (unsigned v = 0; v < nvw; ++v)
				__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(buf + v * 1024u), 16,
				                                         (int)(v * 1024u + lane * 16u), 0, 0, 0);
		}
		else
		{
#pragma unroll
			for (int v = 0; v < NVW; ++v)
				__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(buf + v * 1024u), 16,
				                                         (int)(v * 1024u + lane * 16u), 0, 0, 0);
		}
		return (unsigned)__builtin_amdgcn_readfirstlane((int)(shift | (frames << 16)));
	};

	// the first wave-tile's window is fetched BEFORE the rows are staged: its round trip runs under theirs
	unsigned first_info = 0;
	if (global_wave < n_chunks)
	{
		const uint64_t first = global_wave << chunk_shift;
		const unsigned n = (unsigned)((a.n_out - first < WT) ? (a.n_out - first) : WT);
		first_info = fetch(first, n, my_buf);
	}

	// stage the polyphase rows once per workgroup (as magnitudes where the slot signs are fixed): the only barrier of the kernel
	{
		const u32x4 *src = reinterpret_cast<const u32x4 *>(a.d_rows);
		u32x4 *dst = reinterpret_cast<u32x4 *>(smem);
		// rows land SWIZZLED within their block of 16 (see one_frame2): the global image is the plain one, shared by plans of every
		// increment; the multiplier that suits THIS increment is the plan's (host: cr_poly_pick_swizzle)
		auto place = [&](unsigned r) { return (r & ~15u) | ((__umul24(r >> 4, a.swizzle) + r) & 15u); };
		auto staged = [&](u32x4 v, int q) {
			int e[4] = {(int)v.x, (int)v.y, (int)v.z, (int)v.w};
			if constexpr (!SIGNED)
			{
#pragma unroll
				for (int k = 0; k < 4; ++k)
				{
					const int slot = 4 * q + k;
					if (slot < TT && ((NEGMASK >> slot) & 1u))
						e[k] = -e[k];
					if (MOVARM && slot < TT && !((SAFEMASK >> slot) & 1u))
						e[k] = (int)((unsigned)e[k] << 15);
				}
			}
			u32x4 w;
			w.x = (unsigned)e[0];
			w.y = (unsigned)e[1];
			w.z = (unsigned)e[2];
			w.w = (unsigned)e[3];
			return w;
		};
		if constexpr (RT)
		{
			for (unsigned q = 0; q < planes_total; ++q)
				for (unsigned r = tid; r < a.plane_rows; r += NTHREADS)
					dst[q * a.plane_rows + place(r)] = staged(src[q * a.plane_rows + r], 0);
		}
		else
		{
			// every plane's load of a row before any store: one round trip per trip of this loop instead of one per plane
			for (unsigned r = tid; r < a.plane_rows; r += NTHREADS)
			{
				u32x4 v[RS / 4];
#pragma unroll
				for (int q = 0; q < RS / 4; ++q)
					v[q] = src[(unsigned)q * a.plane_rows + r];
				const unsigned at = place(r);
#pragma unroll
				for (int q = 0; q < RS / 4; ++q)
					dst[(unsigned)q * a.plane_rows + at] = staged(v[q], q);
			}
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the rows' loads are hipcc's; this is for the window in flight)
	__syncthreads();

	// packed window -> X = sample << 16, one dword per sample, in window order (the frames beyond the caller's buffer were delivered
	// as zeros).  The window is converted dword by dword from the aligned dword its first sample sits in: for an odd channel
	// count that sample may be the dword's HIGH half, and sample j of the window is then X[1 + j] (returned: 0 or 1).
	auto expand = [&](const unsigned char *buf, unsigned shift_frames) -> unsigned {
		if constexpr (DUAL)
		{
			// two packed mono windows -> interleaved entries {X of the first window's sample p, X of the second's}: entry p + 1 (a
			// window that starts in the high half of its first dword has a "sample -1" in front).  A lane converts dword k of each
			// window: two samples each, two entries each, written as a pair of dwords two apart.
			const unsigned shift_a = shift_frames & 0xFFu, shift_b = (shift_frames >> 8) & 0xFFu, frames = shift_frames >> 16;
			const unsigned odd_a = (shift_a >> 1) & 1u, odd_b = (shift_b >> 1) & 1u;
			const unsigned dwords = (1u + frames + 1u) / 2u;   // wave-uniform: enough for either window
			const int *from_a = reinterpret_cast<const int *>(buf + (shift_a & ~3u)) + lane;
			const int *from_b = reinterpret_cast<const int *>(buf + BUF / 2u + (shift_b & ~3u)) + lane;
			int *x = reinterpret_cast<int *>(my_x);
#pragma unroll
			for (unsigned k = 0; k < (unsigned)NVW * 2u; ++k)   // (a window is half a buffer: NVW * 2 trips of 64 dwords)
			{
				if (k * 64u >= dwords)
					break;
				const unsigned d = k * 64u + lane;
				if (d < dwords)
				{
					const int fa = from_a[k * 64u], fb = from_b[k * 64u];
					int a0, a1, b0, b1;
					if constexpr (MOVARM)
					{
						a0 = (int)((unsigned)fa << 16) >> 15;
						a1 = (fa >> 16) * 2;
						b0 = (int)((unsigned)fb << 16) >> 15;
						b1 = (fb >> 16) * 2;
					}
					else
					{
						a0 = (int)((unsigned)fa << 16);
						a1 = (int)((unsigned)fa & 0xFFFF0000u);
						b0 = (int)((unsigned)fb << 16);
						b1 = (int)((unsigned)fb & 0xFFFF0000u);
					}
					// sample p = 2 d - odd (and p + 1) -> entries p + 1, p + 2: dwords 2 (p + 1) [+ 1 for the second window], two further on
					int *ea = x + 2u * (2u * d - odd_a + 1u), *eb = x + 2u * (2u * d - odd_b + 1u) + 1u;
					ea[0] = a0;
					ea[2] = a1;
					eb[0] = b0;
					eb[2] = b1;
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			return 2u;   // (entry p + 1: one entry - two dwords - in front, whatever the windows' alignment)
		}
		const unsigned shift = shift_frames & 0xFFFFu, frames = shift_frames >> 16;
		const unsigned odd = (shift >> 1) & 1u;
		const unsigned dwords = (odd + frames * CH + 1u) / 2u;   // wave-uniform
		// straight-line over the whole buffer, left early on a scalar compare: two VALU per 64 dwords (the lanes beyond the window
		// convert whatever the buffer holds - at most 12 bytes of it the neighbouring buffer's - into X entries nobody reads)
		const int *from = reinterpret_cast<const int *>(buf + (shift & ~3u)) + lane;
		i32x2 *to = reinterpret_cast<i32x2 *>(my_x) + lane;
		auto convert = [&](unsigned k) {
			const int f = from[k * 64u];
			i32x2 x;
			if constexpr (MOVARM)
			{
				x.x = (int)((unsigned)f << 16) >> 15;   // 2 * sample
				x.y = (f >> 16) * 2;
			}
			else
			{
				x.x = (int)((unsigned)f << 16);
				x.y = (int)((unsigned)f & 0xFFFF0000u);
			}
			to[k * 64u] = x;
		};
		if constexpr (RT)
		{
			for (unsigned k = 0; k * 64u < dwords; ++k)
				convert(k);
		}
		else
		{
#pragma unroll
			for (unsigned k = 0; k < (unsigned)NVW * 4u; ++k)
			{
				if (k * 64u >= dwords)
					break;
				convert(k);
			}
		}
		// same wave: its LDS operations complete in order
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		return odd;
	};

		// accumulator pairs pinned to physical registers (see k_up2): A+ v[120:121], B+ v[122:123], A- v[124:125], B- v[126:127]
		// (an accumulator's FIRST tap adds to {arm, 0} - a scratch register paired with a register that holds 0 for the whole
		// kernel, v[118:119] - instead of to itself: no v_mov to clear it)
#define CRHIP_W2_TAP_FIRST(LO, HI, VLO, VHI, X, W)                                                                                 \
	asm("v_ashrrev_i32_e32 v118, 31, %2\n\t"                                                                                     \
	    "v_mad_i64_i32 v[" #LO ":" #HI "], vcc, %2, %3, v[118:119]"                                                               \
	    : "=&{v" #LO "}"(VLO), "=&{v" #HI "}"(VHI) : "v"(X), "v"(W), "{v119}"(zero119) : "vcc", "v118")
// (the arming move is the COMPILER's instruction: between two asm statements that touch a common pinned register hipcc pads with an
// s_nop unless an instruction of its own stands in between - 28 s_nop per 15-slot stereo frame with the move inside the statement, 7 so)
#define CRHIP_W2_TAP_MOV(LO, HI, VLO, VHI, X, W)                                                                                   \
	VLO = (X);                                                                                                                     \
	asm("v_mad_i64_i32 v[" #LO ":" #HI "], vcc, %2, %3, v[" #LO ":" #HI "]"                                                      \
	    : "+{v" #LO "}"(VLO), "+{v" #HI "}"(VHI) : "v"(X), "v"(W) : "vcc")
#define CRHIP_W2_TAP_MOV_SAFE(LO, HI, VLO, VHI, X, W)                                                                              \
	asm("v_lshlrev_b32_e32 v118, 15, %2\n\t"                                                                                     \
	    "v_mov_b32_e32 v" #LO ", %2\n\t"                                                                                         \
	    "v_mad_i64_i32 v[" #LO ":" #HI "], vcc, v118, %3, v[" #LO ":" #HI "]"                                                    \
	    : "=&{v" #LO "}"(VLO), "+{v" #HI "}"(VHI) : "v"(X), "v"(W) : "vcc", "v118")
#define CRHIP_W2_TAP_MOV_FIRST(LO, HI, VLO, VHI, X, W)                                                                             \
	asm("v_mov_b32_e32 v118, %2\n\t"                                                                                             \
	    "v_mad_i64_i32 v[" #LO ":" #HI "], vcc, %2, %3, v[118:119]"                                                               \
	    : "=&{v" #LO "}"(VLO), "=&{v" #HI "}"(VHI) : "v"(X), "v"(W), "{v119}"(zero119) : "vcc", "v118")
#define CRHIP_W2_TAP_SIGNED_FIRST(LO, HI, VLO, VHI, X, W)                                                                          \
	asm("v_xor_b32_sdwa v118, sext(%2), sext(%3) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3\n\t"      \
	    "v_mad_i64_i32 v[" #LO ":" #HI "], vcc, %2, %3, v[118:119]"                                                               \
	    : "=&{v" #LO "}"(VLO), "=&{v" #HI "}"(VHI) : "v"(X), "v"(W), "{v119}"(zero119) : "vcc", "v118")
#define CRHIP_W2_TAP(LO, HI, VLO, VHI, X, W)                                                                                       \
	asm("v_ashrrev_i32_e32 v" #LO ", 31, %2\n\t"                                                                                  \
	    "v_mad_i64_i32 v[" #LO ":" #HI "], vcc, %2, %3, v[" #LO ":" #HI "]"                                                      \
	    : "=&{v" #LO "}"(VLO), "+{v" #HI "}"(VHI) : "v"(X), "v"(W) : "vcc")
#define CRHIP_W2_TAP_SIGNED(LO, HI, VLO, VHI, X, W)                                                                                \
	asm("v_xor_b32_sdwa v" #LO ", sext(%2), sext(%3) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3\n\t"   \
	    "v_mad_i64_i32 v[" #LO ":" #HI "], vcc, %2, %3, v[" #LO ":" #HI "]"                                                      \
	    : "=&{v" #LO "}"(VLO), "+{v" #HI "}"(VHI) : "v"(X), "v"(W) : "vcc")
	constexpr int FIRST_POS = SIGNED ? 0 : __builtin_ctz(~NEGMASK), FIRST_NEG = SIGNED ? -1 : (NEGMASK ? __builtin_ctz(NEGMASK) : -1);
	// the taps of slot s on the pinned accumulators: channel(s) xa_s (and xb_s when `pair`) times w_s.  Inlined into fully unrolled
	// loops: s and pair are constants wherever this is called.
	auto slot_taps = [&](int s, bool pair, int xa_s, int xb_s, int w_s, int &lo0, int &hi0, int &lo1, int &hi1, int &lo2, int &hi2, int &lo3, int &hi3) __attribute__((always_inline)) {
		if constexpr (SIGNED)
		{
			if (s == FIRST_POS)
			{
				CRHIP_W2_TAP_SIGNED_FIRST(120, 121, lo0, hi0, xa_s, w_s);
				if (pair)
					CRHIP_W2_TAP_SIGNED_FIRST(122, 123, lo1, hi1, xb_s, w_s);
			}
			else
			{
				CRHIP_W2_TAP_SIGNED(120, 121, lo0, hi0, xa_s, w_s);
				if (pair)
					CRHIP_W2_TAP_SIGNED(122, 123, lo1, hi1, xb_s, w_s);
			}
		}
		else if constexpr (MOVARM)
		{
			static_assert(!MOVARM || (!((SAFEMASK >> (FIRST_POS < 0 ? 0 : FIRST_POS)) & 1u) && !((SAFEMASK >> (FIRST_NEG < 0 ? 0 : FIRST_NEG)) & 1u) && !(SAFEMASK & NEGMASK)),
			              "an accumulator's first tap is an ordinary slot, and the unrestricted slots are positive ones");
			if ((NEGMASK >> s) & 1u)
			{
				if (s == FIRST_NEG)
				{
					CRHIP_W2_TAP_MOV_FIRST(124, 125, lo2, hi2, xa_s, w_s);
					if (pair)
						CRHIP_W2_TAP_MOV_FIRST(126, 127, lo3, hi3, xb_s, w_s);
				}
				else
				{
					CRHIP_W2_TAP_MOV(124, 125, lo2, hi2, xa_s, w_s);
					if (pair)
						CRHIP_W2_TAP_MOV(126, 127, lo3, hi3, xb_s, w_s);
				}
			}
			else if (s == FIRST_POS)
			{
				CRHIP_W2_TAP_MOV_FIRST(120, 121, lo0, hi0, xa_s, w_s);
				if (pair)
					CRHIP_W2_TAP_MOV_FIRST(122, 123, lo1, hi1, xb_s, w_s);
			}
			else if ((SAFEMASK >> s) & 1u)
			{
				CRHIP_W2_TAP_MOV_SAFE(120, 121, lo0, hi0, xa_s, w_s);
				if (pair)
					CRHIP_W2_TAP_MOV_SAFE(122, 123, lo1, hi1, xb_s, w_s);
			}
			else
			{
				CRHIP_W2_TAP_MOV(120, 121, lo0, hi0, xa_s, w_s);
				if (pair)
					CRHIP_W2_TAP_MOV(122, 123, lo1, hi1, xb_s, w_s);
			}
		}
		else if ((NEGMASK >> s) & 1u)
		{
			if (s == FIRST_NEG)
			{
				CRHIP_W2_TAP_FIRST(124, 125, lo2, hi2, xa_s, w_s);
				if (pair)
					CRHIP_W2_TAP_FIRST(126, 127, lo3, hi3, xb_s, w_s);
			}
			else
			{
				CRHIP_W2_TAP(124, 125, lo2, hi2, xa_s, w_s);
				if (pair)
					CRHIP_W2_TAP(126, 127, lo3, hi3, xb_s, w_s);
			}
		}
		else
		{
			if (s == FIRST_POS)
			{
				CRHIP_W2_TAP_FIRST(120, 121, lo0, hi0, xa_s, w_s);
				if (pair)
					CRHIP_W2_TAP_FIRST(122, 123, lo1, hi1, xb_s, w_s);
			}
			else
			{
				CRHIP_W2_TAP(120, 121, lo0, hi0, xa_s, w_s);
				if (pair)
					CRHIP_W2_TAP(122, 123, lo1, hi1, xb_s, w_s);
			}
		}
	};
#undef CRHIP_W2_TAP
#undef CRHIP_W2_TAP_SIGNED
#undef CRHIP_W2_TAP_FIRST
#undef CRHIP_W2_TAP_MOV
#undef CRHIP_W2_TAP_MOV_SAFE
#undef CRHIP_W2_TAP_MOV_FIRST
#undef CRHIP_W2_TAP_SIGNED_FIRST

	// one output frame from the expanded window: CH normalised results into out[]
	auto one_frame2 = [&](unsigned rel, unsigned x_odd, int *out) {
		unsigned shift;
		const unsigned row = row_of<MODE>(a, rel & 0xFFFFu, shift);
		// The lanes of a wave hold consecutive output frames, so their rows step by a fixed amount (83.2 rows per lane at
		// 44.1 -> 48 kHz) and the 16 lanes a ds_read_b128 services together fall on 5-8 of the 16 bank slots: the row reads were
		// half of this kernel's LDS cycles as conflicts (profiles/r02_kwave2_trials.log).  Within each block of 16 rows the rows
		// are rotated by a host-chosen multiple of the block number, which spreads them over all 16 slots.
		const unsigned phys = (ABL & 2) ? 0u : ((row & ~15u) | ((__umul24(row >> 4, a.swizzle) + row) & 15u));
		// LDS byte address of sample 0 of slot 0 (the low 32 bits of a __shared__ pointer are the LDS address)
		const unsigned win_at = (unsigned)(uintptr_t)my_x + (x_odd + ((ABL & 1) ? 0u : ((rel >> 16) + shift)) * CH) * 4u;
		// LDS byte address of this frame's row in plane 0
		const unsigned row_at = (unsigned)(uintptr_t)smem + phys * 16u;
		const unsigned plane_bytes = a.plane_rows * 16u;

		if constexpr (RT)
		{
			// Run-time slot count: one row plane per trip - a ds_read_b128 of four weights, the window samples of its four slots for
			// every channel, the taps.  A padded slot has weight 0 and adds exactly 0 whatever its window read returns.  One
			// accumulator pair per channel, pinned to v[128 - 2 CH .. 127].
			const unsigned weight_planes = planes_total - 1u;
			int hi[CH], lo[CH];
#pragma unroll
			for (int c = 0; c < CH; ++c)
				hi[c] = 0;
			unsigned row_q = row_at, win_q = win_at;
#define CRHIP_W2_TAPC(LO, HI, VLO, VHI, X, W)                                                                                      \
	asm("v_xor_b32_sdwa v" #LO ", sext(%2), sext(%3) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3\n\t"   \
	    "v_mad_i64_i32 v[" #LO ":" #HI "], vcc, %2, %3, v[" #LO ":" #HI "]"                                                      \
	    : "=&{v" #LO "}"(VLO), "+{v" #HI "}"(VHI) : "v"(X), "v"(W) : "vcc")
			auto tap = [&](auto pair_tag, int &vlo, int &vhi, int x, int w) {
				constexpr int P = decltype(pair_tag)::value;   // pair P of the eight: v[112 + 2 P : 113 + 2 P]
				if constexpr (P == 0) CRHIP_W2_TAPC(112, 113, vlo, vhi, x, w);
				else if constexpr (P == 1) CRHIP_W2_TAPC(114, 115, vlo, vhi, x, w);
				else if constexpr (P == 2) CRHIP_W2_TAPC(116, 117, vlo, vhi, x, w);
				else if constexpr (P == 3) CRHIP_W2_TAPC(118, 119, vlo, vhi, x, w);
				else if constexpr (P == 4) CRHIP_W2_TAPC(120, 121, vlo, vhi, x, w);
				else if constexpr (P == 5) CRHIP_W2_TAPC(122, 123, vlo, vhi, x, w);
				else if constexpr (P == 6) CRHIP_W2_TAPC(124, 125, vlo, vhi, x, w);
				else CRHIP_W2_TAPC(126, 127, vlo, vhi, x, w);
			};
			constexpr bool EVEN = CH % 2 == 0;
			for (unsigned q = 0; q < weight_planes; ++q)
			{
				i32x4 wq;
				int x[4][CH];
				i32x2 xv[4][(CH + 1) / 2];
				asm volatile("ds_read_b128 %0, %1" : "=v"(wq) : "v"(row_q));
#pragma unroll
				for (int s = 0; s < 4; ++s)
				{
#pragma unroll
					for (int c = 0; c < CH; c += (EVEN ? 2 : 1))
					{
						if constexpr (EVEN)
							asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xv[s][c / 2]) : "v"(win_q), "n"((s * CH + c) * 4));
						else
							asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x[s][c]) : "v"(win_q), "n"((s * CH + c) * 4));
					}
				}
				// the taps of slot s start when ITS reads have landed (in order: a counted wait), the later slots' under them
				constexpr int READS_PER_SLOT = EVEN ? CH / 2 : CH;
				asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(3 * READS_PER_SLOT < 15 ? 3 * READS_PER_SLOT : 15) : "memory");
				asm volatile("" : "+v"(wq));
				const int wv[4] = {wq.x, wq.y, wq.z, wq.w};
#pragma unroll
				for (int s = 0; s < 4; ++s)
				{
					if (s > 0)
						asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((3 - s) * READS_PER_SLOT < 15 ? (3 - s) * READS_PER_SLOT : 15) : "memory");
#pragma unroll
					for (int c = 0; c < CH; ++c)
					{
						if constexpr (EVEN)
						{
							if (c % 2 == 0)
							{
								asm volatile("" : "+v"(xv[s][c / 2]));
								x[s][c] = xv[s][c / 2].x;
								x[s][c + 1] = xv[s][c / 2].y;
							}
						}
						else
							asm volatile("" : "+v"(x[s][c]));
					}
#pragma unroll
					for (int c = 0; c < CH; ++c)
					{
						switch (8 - CH + c)
						{
							case 0: tap(std::integral_constant<int, 0>(), lo[c], hi[c], x[s][c], wv[s]); break;
							case 1: tap(std::integral_constant<int, 1>(), lo[c], hi[c], x[s][c], wv[s]); break;
							case 2: tap(std::integral_constant<int, 2>(), lo[c], hi[c], x[s][c], wv[s]); break;
							case 3: tap(std::integral_constant<int, 3>(), lo[c], hi[c], x[s][c], wv[s]); break;
							case 4: tap(std::integral_constant<int, 4>(), lo[c], hi[c], x[s][c], wv[s]); break;
							case 5: tap(std::integral_constant<int, 5>(), lo[c], hi[c], x[s][c], wv[s]); break;
							case 6: tap(std::integral_constant<int, 6>(), lo[c], hi[c], x[s][c], wv[s]); break;
							default: tap(std::integral_constant<int, 7>(), lo[c], hi[c], x[s][c], wv[s]); break;
						}
					}
				}
				row_q += plane_bytes;
				win_q += 16u * CH;
			}
#undef CRHIP_W2_TAPC
			// (row_q now points at this row's entry in the plane after the weights: the reciprocal)
			int reciprocal;
			asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(reciprocal) : "v"(row_q) : "memory");
#pragma unroll
			for (int c = 0; c < CH; ++c)
			{
				(void)lo[c];
				// (acc * reciprocal) / 32768 toward zero in 64 bits: right for either range the host may have classified the rows as
				const long long v = (long long)hi[c] * (long long)reciprocal + (long long)((unsigned)(hi[c] >> 31) >> 17);
				out[c] = (int)(v >> 15);
			}
			return;
		}

		constexpr int NQ = RS / 4;
		int w[RS];
		i32x4 wv[NQ];   // (the row reads land here)

		// channels two at a time (the last one alone when CH is odd).  Every LDS read of the frame is issued up front - row plane q,
		// then the window samples of its four slots, plane by plane - and the taps of plane q start as soon as ITS reads have
		// landed (LDS returns in order: a counted lgkmcnt), the rest of the reads completing underneath the arithmetic.  All of
		// them are inline assembly so that the counts are ours: hipcc's own waits only know the reads it issued itself.  Nothing
		// may touch a destination register between its read and the "+v" statement behind the wait that covers it.
		static_for<(CH + 1) / 2>([&](auto c_tag) {
			constexpr int c0 = 2 * decltype(c_tag)::value;
			constexpr bool EVEN = CH % 2 == 0;
			constexpr bool pair = c0 + 1 < CH;
			constexpr int reads_per_slot = (pair && !EVEN) ? 2 : 1;
			// The window as separate reads: left to itself hipcc pairs them into ds_read2_b64, which moves 128 B per clock where
			// ds_read_b64 moves 256 (MI355X_MICROARCH.md, LDS) - and this kernel is as much LDS- as VALU-bound.  8-byte reads need
			// 8-byte alignment: even channel counts only.
			constexpr int TTN = TT > 0 ? TT : 1;   // (this part is never reached by the run-time-slot form; it still has to compile)
			int xa[TTN], xb[TTN];
			i32x2 xv[TTN];   // (the 8-byte reads land here)
			// The reads of plane q - its row vector (first channel pair only) and the window samples of its four slots - as one unit.
			// They are NOT all issued up front any more: lgkmcnt counts to 15, so with a frame's 42 reads (33 slots) in flight the
			// first taps waited for 27 of them, and the waves of a CU fell into step - all reading, then all multiplying (LDS 56 %
			// busy beside a VALU 47 % busy, profiles/r03_dn8_before_pmc_summary.txt).  Now AHEAD planes are in flight (at most 15
			// reads), the taps of plane q start when ITS reads have landed, and plane q + AHEAD is requested right behind them: a
			// wave asks the LDS for data at the pace it consumes it.
			auto issue_plane = [&](auto q_tag) {
				constexpr int q = decltype(q_tag)::value;
				if constexpr (c0 == 0)
				{
					if constexpr (MODE == CRHIP_ROWMODE_UPSAMPLE)
					{
						// pure upsampling has ONE row image whatever the ratio (UP_PLANE_ROWS rows per plane, checked at launch): the
						// further planes are immediate offsets
						asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wv[q]) : "v"(row_at), "n"(q * (int)UP_PLANE_ROWS * 16));
					}
					else
					{
						const unsigned at = row_at + (unsigned)q * plane_bytes;
						asm volatile("ds_read_b128 %0, %1" : "=v"(wv[q]) : "v"(at));
					}
				}
				// (four slots, written out: asm operands inside a further nested generic lambda do not capture for clang)
#define CRHIP_W2_READ_SLOT(K)                                                                                                      \
	if constexpr (4 * q + K < TT)                                                                                                  \
	{                                                                                                                              \
		constexpr int s = 4 * q + K;                                                                                               \
		if constexpr (pair && EVEN)                                                                                                \
			asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xv[s]) : "v"(win_at), "n"((s * CH + c0) * 4));                      \
		else                                                                                                                       \
		{                                                                                                                          \
			asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(xa[s]) : "v"(win_at), "n"((s * CH + c0) * 4));                      \
			if constexpr (pair)                                                                                                    \
				asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(xb[s]) : "v"(win_at), "n"((s * CH + c0 + 1) * 4));              \
		}                                                                                                                          \
	}
				CRHIP_W2_READ_SLOT(0)
				CRHIP_W2_READ_SLOT(1)
				CRHIP_W2_READ_SLOT(2)
				CRHIP_W2_READ_SLOT(3)
#undef CRHIP_W2_READ_SLOT
			};
			// planes in flight: as many as 15 outstanding reads allow (a plane is a row vector + up to 4 x reads_per_slot reads)
			constexpr int plane_reads_max = (c0 == 0 ? 1 : 0) + 4 * reads_per_slot;
			constexpr int AHEAD_RAW = 15 / plane_reads_max;
			constexpr int AHEAD = AHEAD_RAW < 1 ? 1 : (AHEAD_RAW > NQ ? NQ : AHEAD_RAW);
			static_for<AHEAD>([&](auto q_tag) { issue_plane(q_tag); });

			// accumulator pairs pinned to physical registers (see k_up2): A+ v[120:121], B+ v[122:123], A- v[124:125], B- v[126:127]
			// (an accumulator's FIRST tap adds to {arm, 0} - a scratch register paired with a register that holds 0 for the whole
			// kernel, v[118:119] - instead of to itself: no v_mov to clear it)
			int lo0, hi0 = 0, lo1, hi1 = 0, lo2, hi2 = 0, lo3, hi3 = 0;
			static_for<NQ>([&](auto q_tag) {
				constexpr int q = decltype(q_tag)::value;
				// reads issued after the last one of plane q - the planes ahead of it - may stay in flight
				constexpr int later = wave2_reads_ahead(c0 == 0, reads_per_slot, TT, q, AHEAD, NQ);
				static_assert(later <= 15, "lgkmcnt counts to 15");
				asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(later) : "memory");
				// (the plane that keeps AHEAD of them in flight, before the taps: its round trip runs under them)
				if constexpr (q + AHEAD < NQ)
					issue_plane(std::integral_constant<int, q + AHEAD>());
				if (c0 == 0)
				{
					asm volatile("" : "+v"(wv[q]));
					w[4 * q] = wv[q].x;
					w[4 * q + 1] = wv[q].y;
					w[4 * q + 2] = wv[q].z;
					w[4 * q + 3] = wv[q].w;
				}
#pragma unroll
				for (int s = 4 * q; s < 4 * q + 4 && s < TT; ++s)
				{
					if (pair && EVEN)
					{
						asm volatile("" : "+v"(xv[s]));
						xa[s] = xv[s].x;
						xb[s] = xv[s].y;
					}
					else if (pair)
						asm volatile("" : "+v"(xa[s]), "+v"(xb[s]));
					else
					{
						asm volatile("" : "+v"(xa[s]));
						xb[s] = 0;
					}
					slot_taps(s, pair, xa[s], xb[s], w[s], lo0, hi0, lo1, hi1, lo2, hi2, lo3, hi3);
				}
			});
			(void)lo0;
			(void)lo1;
			(void)lo2;
			(void)lo3;
			const int acc0 = SIGNED ? hi0 : hi0 - hi2;
			const int acc1 = SIGNED ? hi1 : hi1 - hi3;
			if constexpr (NORM == CRHIP_NORM_U32)
			{
				const long long v0 = (long long)acc0 * (long long)w[TT] + (long long)((unsigned)(acc0 >> 31) >> 17);
				out[c0] = (int)(v0 >> 15);
				if (pair)
				{
					const long long v1 = (long long)acc1 * (long long)w[TT] + (long long)((unsigned)(acc1 >> 31) >> 17);
					out[c0 + 1] = (int)(v1 >> 15);
				}
			}
			else
			{
				out[c0] = normalise<NORM>(acc0, w[TT]);
				if (pair)
					out[c0 + 1] = normalise<NORM>(acc1, w[TT]);
			}
		});
	};

	// dual mono: one descriptor over the mono output, from its first frame to the end of the second half of the stream (H + the second
	// half's frames): frame j goes to [j], its partner to [j + H] - beyond the range where it does not exist, and the store is dropped
	const uint64_t dual_base = reinterpret_cast<uint64_t>(a.d_out);
	const uint64_t dual_bytes = ((uint64_t)a.dual_out_frames + a.dual_valid_frames) * 4u;
	const __amdgpu_buffer_rsrc_t dual_rsrc = __builtin_amdgcn_make_buffer_rsrc(
	    reinterpret_cast<void *>(((uint64_t)__builtin_amdgcn_readfirstlane((int)(unsigned)(dual_base >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)dual_base)), 0,
	    (int)__builtin_amdgcn_readfirstlane((int)(unsigned)(!DUAL ? 0u : (dual_bytes > 0xFFFFFFFCull ? 0xFFFFFFFCull : dual_bytes))), 0x00020000);
	const unsigned dual_second = __builtin_amdgcn_readfirstlane(a.dual_out_frames * 4u);
	auto store_frame = [&](uint64_t frame, const int *out) {
		if constexpr (DUAL)
		{
			const unsigned at = (unsigned)frame * 4u;   // (frame < H < 2^30: the host)
			__builtin_amdgcn_raw_buffer_store_b32(out[0], dual_rsrc, (int)at, 0, NT ? 2 : 0);
			__builtin_amdgcn_raw_buffer_store_b32(out[1], dual_rsrc, (int)at, (int)dual_second, NT ? 2 : 0);
			return;
		}
		if constexpr (OUT16)
			store_shorts<CH, NT>(reinterpret_cast<short *>(a.d_out) + frame * CH, out);
		else
			store_ints<CH, NT>(reinterpret_cast<int *>(a.d_out) + frame * CH, out);
	};


	if (global_wave >= n_chunks)
	{
		retire();
		return;
	}

	uint64_t chunk = global_wave;
	unsigned cur = 0, packed_info = first_info;   // (fetched before the rows were staged, landed before the barrier)

	// One chunk: its wave-tiles, the window of the following chunk's first tile fetched under the last one.  `next_of` names that
	// chunk (or ~0) when asked, in the chunk's last tile.  Returns the chunk to go on with, ~0 when the wave is done.
	auto run_chunk = [&](uint64_t this_chunk, auto next_of) -> uint64_t {
		uint64_t next_chunk = ~0ull;
		const uint64_t chunk_first = this_chunk << chunk_shift;

		for (unsigned j = 0; j < CW; ++j)
		{
			const uint64_t first = chunk_first + (uint64_t)j * WT;
			const unsigned n = (unsigned)((a.n_out - first < WT) ? (a.n_out - first) : WT);
			const bool last_of_stream = first + n >= a.n_out;
			unsigned next_info = 0;
			bool have_next = false;

			// this wave-tile's window becomes X (the previous tile's frames have all been read: same wave, in order) ...
			const unsigned x_odd = expand(my_buf + cur * BUF, packed_info);

			// ... and the DMA of the wave-tile after it starts (the other packed buffer was expanded one step ago)
			if (!last_of_stream && j + 1 < CW)
			{
				const uint64_t nf = first + WT;
				const unsigned nn = (unsigned)((a.n_out - nf < WT) ? (a.n_out - nf) : WT);
				next_info = fetch(nf, nn, my_buf + (cur ^ 1u) * BUF);
				have_next = true;
			}
			else
			{
				// (the stream's last chunk is the last of its wave's sequence, but the question is asked all the same: a ticket
				// drawn for it must have returned before the counters are reset)
				next_chunk = next_of();
				if (next_chunk != ~0ull && !last_of_stream)
				{
					const uint64_t nf = next_chunk << chunk_shift;
					const unsigned nn = (unsigned)((a.n_out - nf < WT) ? (a.n_out - nf) : WT);
					next_info = fetch(nf, nn, my_buf + (cur ^ 1u) * BUF);
					have_next = true;
				}
			}

			const uint64_t pos = a.pos0 + first * (uint64_t)a.increment;
			// Which frame of the 64 a lane takes.  The LDS services a ds_read_b64 in two groups of 32 lanes, conflict-free only if the
			// 32 window bases of a group fall on 32 different frames modulo 32 (stereo: two banks per frame).  With consecutive
			// frames on consecutive lanes that fails wherever 32 steps of the ratio wrap unevenly - 44.1 -> 8 kHz (5.51 frames per
			// lane): 3 of 32 lanes collide and EVERY window read takes 3-4 LDS cycles instead of 2 (43 % of that kernel's LDS cycles
			// were conflicts, profiles/r03_dn8_before_pmc_summary.txt).  lane_map 1 gives the lower 32 lanes the even frames and
			// the upper 32 the odd ones: 11.02 frames per lane, an odd step, 32 different residues.  The host picks per plan
			// (cr_context.c plan_pick_lane_map: a model of those groups); the stores of a wave still cover the same 64 frames.
			const unsigned flane = a.lane_map ? (((lane & 31u) << 1) | (lane >> 5)) : lane;
			const unsigned lane_rel = __umul24(flane, a.increment) + (unsigned)(pos & 0xFFFFu);
			if (n == WT)
			{
#pragma unroll
				for (int i = 0; i < ITER; ++i)
				{
					int out[CH];
					one_frame2(lane_rel + (unsigned)i * 64u * a.increment, x_odd, out);
					store_frame(first + (unsigned)i * 64u + flane, out);
				}
			}
			else
			{
				for (unsigned jl = flane; jl < n; jl += 64u)
				{
					int out[CH];
					one_frame2(__umul24(jl, a.increment) + (unsigned)(pos & 0xFFFFu), x_odd, out);
					store_frame(first + jl, out);
				}
			}

			if (last_of_stream || !have_next)
				return ~0ull;
			// own DMA landed once only this wave-tile's stores are outstanding (vmcnt is in order)
			if (n == WT)
			{
				if constexpr (ITER * STORES_PER_FRAME <= 63)
					asm volatile("s_waitcnt vmcnt(%0)" ::"i"(ITER * STORES_PER_FRAME) : "memory");
				else
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			}
			else
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			cur ^= 1u;
			packed_info = next_info;
		}
		return next_chunk;
	};

	// the statically dealt rounds ...
	while (chunk != ~0ull && chunk < static_limit)
	{
		const uint64_t this_chunk = chunk;
		chunk = run_chunk(this_chunk, [&]() -> uint64_t {
			const uint64_t next = this_chunk + global_waves;
			return next < n_chunks ? next : ~0ull;
		});
	}
	// ... and the ticketed ones (a ticket is drawn when a chunk is entered and resolved in its last tile)
	while (chunk != ~0ull)
	{
		const unsigned ticket = draw_issue();
		chunk = run_chunk(chunk, [&]() -> uint64_t { return draw_resolve(ticket); });
	}

	retire();
}

} // namespace

#endif // CR_KWAVE2_HPP
