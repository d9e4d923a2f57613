// cr_device.hpp - device-side building blocks shared by the kernels of libclownresampler_amd (gfx950 only): the fixed-point tap
// arithmetic in its SDWA / 64-bit-chain forms, packed frames, the polyphase row index, one output frame from LDS, stores.
// Included by every kernel header (cr_kpoly.hpp, cr_kwave.hpp, cr_kup.hpp); everything has internal linkage.
#ifndef CR_DEVICE_HPP
#define CR_DEVICE_HPP

#include <hip/hip_runtime.h>

#include <stdint.h>
#include <string.h>

#include <type_traits>
#include <utility>

#include "crhip.h"

namespace
{

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

// compile-time loop: body(std::integral_constant<int, i>) for i in [0, N) - every index a constant expression, whatever hipcc's
// unrolling thresholds think of a 432-tap body (left to `#pragma unroll` the accumulator arrays went to scratch)
template <int N, typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&body, std::integer_sequence<int, I...>)
{
	(body(std::integral_constant<int, I>()), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&body)
{
	static_for_impl<N>(static_cast<F &&>(body), std::make_integer_sequence<int, N>());
}

// ---------------------------------------------------------------------------------------------------------
// Fixed-point pieces
// ---------------------------------------------------------------------------------------------------------

// The per-tap term is (sample * weight) / 65536 with C semantics (truncation toward zero), clownresampler.h:1020 via
// :625.  Both operands fit 24 bits and the host only selects the 32-bit kernels for -65536 < weight <= 65536 (cr_plan.c): with
// |sample| <= 2^15 the product then fits int32 (-32768 * 65536 is exactly INT32_MIN), so the low 32 bits the full-rate 24-bit
// multiplier delivers ARE the product; see accumulate_product below.  Larger weights (a caller's own table) go to k_generic.

// (acc * reciprocal) / 32768 with C semantics, clownresampler.h:1033.  Host-proved: |acc| < 2^23,
// 0 < reciprocal < 2^23 and either |acc * reciprocal| < 2^31 (NORM_S31) or < 2^32 (NORM_U32: the product of the
// magnitudes is exact in the low 32 bits of the 24-bit multiplier; truncation toward zero is symmetric in sign).
template <int NORM>
__device__ __forceinline__ int normalise(int acc, int reciprocal)
{
	if constexpr (NORM == CRHIP_NORM_S31)
	{
		// truncation toward zero = + 0x7FFF before the shift when the product is negative; the reciprocal is positive, so
		// that is when the ACCUMULATOR is negative: the bias does not wait for the product and the multiply becomes a
		// multiply-add (4 instructions instead of 5: hipcc otherwise multiplies twice)
		const int bias = (int)((unsigned)(acc >> 31) >> 17);
		return (__mul24(acc, reciprocal) + bias) >> 15;
	}
	else
	{
		const int sign = acc >> 31;
		const unsigned magnitude = (unsigned)((acc ^ sign) - sign);
		const unsigned quotient = __umul24(magnitude, (unsigned)reciprocal) >> 15;
		return ((int)quotient ^ sign) - sign;
	}
}

// ---------------------------------------------------------------------------------------------------------
// Sub-dword (SDWA) forms of the tap arithmetic.  A stereo frame is one dword (left in the low word, right in the
// high word); SDWA operand selects let the multiply read either word sign-extended, and let an add read the high
// word of a register, which IS the shift by 16:
//     x   = v_mul_i32_i24(sext(word k of frame), weight)            product, exact
//     t   = x >> 31                                                  0 / -1
//     x'  = x + (t >>> 16)                 add, src1 = WORD_1 of t   + 0xFFFF when negative (C truncation toward zero)
//     acc = acc + (x' >> 16)               add, src1 = sext(WORD_1 of x')
// 4 VALU per tap and channel instead of the 6-7 the compiler emits for the C expression (it unpacks the words
// separately and redoes the multiply as a mad).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sdwa_add_word1_unsigned(int x, int t)
{
	int r;
	asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(x), "v"(t));
	return r;
}

__device__ __forceinline__ int sdwa_add_word1_signed(int acc, int x)
{
	int r;
	asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(acc), "v"(x));
	return r;
}

// One tap of a packed pair of channels as ONE statement (8 instructions): hipcc pads every asm statement whose outputs
// the next instruction reads with an s_nop, so the four-statement form above costs three pads per tap and channel; here
// the only values that leave the statement are the two accumulators.
__device__ __forceinline__ void sdwa_tap_pair(int &acc_lo, int &acc_hi, int frame, int weight)
{
	int x0, x1, t0, t1;
	asm("v_mul_i32_i24_sdwa %2, sext(%6), %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"
	    "v_mul_i32_i24_sdwa %3, sext(%6), %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
	    "v_ashrrev_i32_e32 %4, 31, %2\n\t"
	    "v_ashrrev_i32_e32 %5, 31, %3\n\t"
	    "v_add_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
	    "v_add_u32_sdwa %3, %3, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
	    "v_add_u32_sdwa %0, %0, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
	    "v_add_u32_sdwa %1, %1, sext(%3) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
	    : "+v"(acc_lo), "+v"(acc_hi), "=&v"(x0), "=&v"(x1), "=&v"(t0), "=&v"(t1)
	    : "v"(frame), "v"(weight));
}

// The same for the FIRST tap of an accumulator pair: the truncated terms are written, not added (no zeroing moves, and
// the final shift is a plain v_ashrrev, which issues at twice the rate of an SDWA add on gfx950).
__device__ __forceinline__ void sdwa_tap_pair_first(int &acc_lo, int &acc_hi, int frame, int weight)
{
	int t0, t1;
	asm("v_mul_i32_i24_sdwa %0, sext(%4), %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"
	    "v_mul_i32_i24_sdwa %1, sext(%4), %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
	    "v_ashrrev_i32_e32 %2, 31, %0\n\t"
	    "v_ashrrev_i32_e32 %3, 31, %1\n\t"
	    "v_add_u32_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
	    "v_add_u32_sdwa %1, %1, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
	    "v_ashrrev_i32_e32 %0, 16, %0\n\t"
	    "v_ashrrev_i32_e32 %1, 16, %1"
	    : "=&v"(acc_lo), "=&v"(acc_hi), "=&v"(t0), "=&v"(t1)
	    : "v"(frame), "v"(weight));
}

__device__ __forceinline__ void sdwa_tap_single_first(int &acc, int sample, int weight)
{
	int t;
	asm("v_mul_i32_i24_e32 %0, %2, %3\n\t"
	    "v_ashrrev_i32_e32 %1, 31, %0\n\t"
	    "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
	    "v_ashrrev_i32_e32 %0, 16, %0"
	    : "=&v"(acc), "=&v"(t)
	    : "v"(sample), "v"(weight));
}

// One tap of one (already sign-extended) sample as one statement (4 instructions).
__device__ __forceinline__ void sdwa_tap_single(int &acc, int sample, int weight)
{
	int x, t;
	asm("v_mul_i32_i24_e32 %1, %3, %4\n\t"
	    "v_ashrrev_i32_e32 %2, 31, %1\n\t"
	    "v_add_u32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
	    "v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
	    : "+v"(acc), "=&v"(x), "=&v"(t)
	    : "v"(sample), "v"(weight));
}

// One tap of the sample in the LOW word of a dword (the odd channel that is left over when a frame is read as dwords).
__device__ __forceinline__ void sdwa_tap_word0(int &acc, int frame, int weight)
{
	int x, t;
	asm("v_mul_i32_i24_sdwa %1, sext(%3), %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"
	    "v_ashrrev_i32_e32 %2, 31, %1\n\t"
	    "v_add_u32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
	    "v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
	    : "+v"(acc), "=&v"(x), "=&v"(t)
	    : "v"(frame), "v"(weight));
}

__device__ __forceinline__ void sdwa_tap_word0_first(int &acc, int frame, int weight)
{
	int t;
	asm("v_mul_i32_i24_sdwa %0, sext(%2), %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"
	    "v_ashrrev_i32_e32 %1, 31, %0\n\t"
	    "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
	    "v_ashrrev_i32_e32 %0, 16, %0"
	    : "=&v"(acc), "=&v"(t)
	    : "v"(frame), "v"(weight));
}

// ... and in the HIGH word (mono: two neighbouring frames of the window share a dword).
__device__ __forceinline__ void sdwa_tap_word1(int &acc, int frame, int weight)
{
	int x, t;
	asm("v_mul_i32_i24_sdwa %1, sext(%3), %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
	    "v_ashrrev_i32_e32 %2, 31, %1\n\t"
	    "v_add_u32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
	    "v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
	    : "+v"(acc), "=&v"(x), "=&v"(t)
	    : "v"(frame), "v"(weight));
}

__device__ __forceinline__ void sdwa_tap_word1_first(int &acc, int frame, int weight)
{
	int t;
	asm("v_mul_i32_i24_sdwa %0, sext(%2), %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
	    "v_ashrrev_i32_e32 %1, 31, %0\n\t"
	    "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
	    "v_ashrrev_i32_e32 %0, 16, %0"
	    : "=&v"(acc), "=&v"(t)
	    : "v"(frame), "v"(weight));
}

// acc += trunc(product / 65536)
template <int ASM>
__device__ __forceinline__ int accumulate_product(int acc, int product)
{
	if constexpr (ASM)
		return sdwa_add_word1_signed(acc, sdwa_add_word1_unsigned(product, product >> 31));
	else
		return acc + ((product + (int)((unsigned)(product >> 31) >> 16)) >> 16);
}

// One input frame from LDS, kept PACKED (two int16 per dword) and multiplied straight out of the dwords by the SDWA
// forms.  An odd channel count leaves one sample over: it sits in the low word of the last dword.
// Frames of an odd channel count start on 2-byte boundaries, and LDS reads that are not naturally aligned are SLOW on
// gfx950: hipcc merges neighbouring 16-bit reads into ds_read_b64 / b32 on 2-byte boundaries, and the mono and 3-channel
// kernels measured 1.5-1.7x slower for it (profiles/).  So an odd frame is read as the ALIGNED dwords that cover it and
// funnel-shifted into place (v_alignbit_b32 by 0 or 16): one instruction per dword, which also replaces the per-sample
// sign extension the unpacked form needed.
template <int CH>
struct Frame
{
	static constexpr bool PACKED = (CH % 2) == 0;
	static constexpr int WORDS = (CH + 1) / 2;
	int v[WORDS];

	__device__ __forceinline__ void load(const unsigned char *p)
	{
		if constexpr (CH == 2)
		{
			v[0] = *reinterpret_cast<const int *>(p);
		}
		else if constexpr (CH == 4)
		{
			const i32x2 d = *reinterpret_cast<const i32x2 *>(p);
			v[0] = d.x;
			v[1] = d.y;
		}
		else if constexpr (CH == 8)
		{
			const i32x4 d = *reinterpret_cast<const i32x4 *>(p);
			v[0] = d.x;
			v[1] = d.y;
			v[2] = d.z;
			v[3] = d.w;
		}
		else if constexpr (PACKED)
		{
#pragma unroll
			for (int k = 0; k < WORDS; ++k)
				v[k] = reinterpret_cast<const int *>(p)[k];
		}
		else
		{
			const unsigned odd = (unsigned)reinterpret_cast<uintptr_t>(p) & 2u;   // the frame starts in the high half of a dword
			const unsigned *q = reinterpret_cast<const unsigned *>(p - odd);
			unsigned d[WORDS];
#pragma unroll
			for (int k = 0; k < WORDS; ++k)
				d[k] = q[k];
#pragma unroll
			for (int k = 0; k + 1 < WORDS; ++k)
				v[k] = (int)__builtin_amdgcn_alignbit(d[k + 1], d[k], odd * 8u);
			v[WORDS - 1] = (int)(d[WORDS - 1] >> (odd * 8u));
		}
	}

	// a lane's share in a PADDED tile (padded_frames below): 16 bytes at a 16-byte boundary, the first CH channels its own
	__device__ __forceinline__ void load_padded(const unsigned char *p)
	{
		static_assert(WORDS <= 4, "a lane's share of a padded frame is 16 bytes");
		const i32x4 d = *reinterpret_cast<const i32x4 *>(p);
		const int w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
		for (int k = 0; k < WORDS; ++k)
			v[k] = w[k];
	}

	// the same for a frame that may start on ANY 2-byte boundary whatever its channel count (frames of an odd total channel
	// count shared by two lanes): aligned dwords + funnel shift, as above
	__device__ __forceinline__ void load_any(const unsigned char *p)
	{
		if constexpr (!PACKED)
		{
			load(p);
		}
		else
		{
			const unsigned odd = (unsigned)reinterpret_cast<uintptr_t>(p) & 2u;
			const unsigned *q = reinterpret_cast<const unsigned *>(p - odd);
			unsigned d[WORDS + 1];
#pragma unroll
			for (int k = 0; k < WORDS + 1; ++k)
				d[k] = q[k];
#pragma unroll
			for (int k = 0; k < WORDS; ++k)
				v[k] = (int)__builtin_amdgcn_alignbit(d[k + 1], d[k], odd * 8u);
		}
	}

	// acc = first tap's terms (no previous contents)
	template <int ASM>
	__device__ __forceinline__ void mac_first(int (&acc)[CH], int weight) const
	{
		if constexpr (!ASM)
		{
#pragma unroll
			for (int c = 0; c < CH; ++c)
				acc[c] = 0;
			mac<0>(acc, weight);
		}
		else
		{
#pragma unroll
			for (int k = 0; k < CH / 2; ++k)
				sdwa_tap_pair_first(acc[2 * k], acc[2 * k + 1], v[k], weight);
			if constexpr (!PACKED)
				sdwa_tap_word0_first(acc[CH - 1], v[WORDS - 1], weight);
		}
	}

	template <int ASM>
	__device__ __forceinline__ void mac(int (&acc)[CH], int weight) const
	{
#pragma unroll
		for (int k = 0; k < CH / 2; ++k)
		{
			if constexpr (ASM)
			{
				sdwa_tap_pair(acc[2 * k], acc[2 * k + 1], v[k], weight);
			}
			else
			{
				acc[2 * k] = accumulate_product<0>(acc[2 * k], __mul24((int)(short)v[k], weight));
				acc[2 * k + 1] = accumulate_product<0>(acc[2 * k + 1], __mul24(v[k] >> 16, weight));
			}
		}
		if constexpr (!PACKED)
		{
			if constexpr (ASM)
				sdwa_tap_word0(acc[CH - 1], v[WORDS - 1], weight);
			else
				acc[CH - 1] = accumulate_product<0>(acc[CH - 1], __mul24((int)(short)v[WORDS - 1], weight));
		}
	}
};

// N consecutive MONO frames (int16) starting at p, packed two per dword: pw[k] = frames 2k (low word) and 2k + 1 (high word).
// Aligned dword reads + one funnel shift per dword (see Frame); pw must have (N + 1) / 2 elements.
// (Written as one 16-bit read per frame instead, hipcc merges them into ds_read_b64 / b128 at the window's 2-byte alignment - gfx950
// takes unaligned DS accesses - and saves the six VALU instructions of the alignment; measured TWICE as slow: mono 44.1 -> 48 kHz
// 77 -> 157 us, 48 -> 44.1 kHz 91 -> 134 us, profiles/r02_mono_unaligned_window.log.)
template <int N>
__device__ __forceinline__ void load_mono_window(const unsigned char *p, int *pw)
{
	constexpr int NPW = (N + 1) / 2, NW = (N + 2) / 2;
	const unsigned odd = (unsigned)reinterpret_cast<uintptr_t>(p) & 2u;
	const unsigned *q = reinterpret_cast<const unsigned *>(p - odd);
	unsigned d[NW];
#pragma unroll
	for (int k = 0; k < NW; ++k)
		d[k] = q[k];
#pragma unroll
	for (int k = 0; k < NPW; ++k)
		pw[k] = k + 1 < NW ? (int)__builtin_amdgcn_alignbit(d[k + 1], d[k], odd * 8u) : (int)(d[k] >> (odd * 8u));
}

// One mono tap out of a packed window: frame s of the window, weight w.
template <int ASM, bool FIRST>
__device__ __forceinline__ void mono_tap(int &acc, const int *pw, int s, int weight)
{
	if constexpr (!ASM)
	{
		const int sample = (s & 1) ? (pw[s / 2] >> 16) : (int)(short)pw[s / 2];
		acc = accumulate_product<0>(FIRST ? 0 : acc, __mul24(sample, weight));
	}
	else if (s & 1)
	{
		if constexpr (FIRST)
			sdwa_tap_word1_first(acc, pw[s / 2], weight);
		else
			sdwa_tap_word1(acc, pw[s / 2], weight);
	}
	else
	{
		if constexpr (FIRST)
			sdwa_tap_word0_first(acc, pw[s / 2], weight);
		else
			sdwa_tap_word0(acc, pw[s / 2], weight);
	}
}

// The frames of a tap window for an ODD channel count above one.  A frame is CH * 2 = 2 (mod 4) bytes, so consecutive frames
// alternate between starting on a dword and in the middle of one; both aligned bases and both funnel shifts are formed once
// per window, and every frame is then read at an immediate offset.
template <int CH, int FB>
struct OddWindow
{
	const unsigned char *even_base, *odd_base;   // aligned base of frame s is {even,odd}_base + s * FB for even / odd s
	unsigned even_shift, odd_shift;

	__device__ __forceinline__ explicit OddWindow(const unsigned char *p)
	{
		static_assert(CH % 2 == 1 && FB % 4 == 2, "frames of an odd channel count, one lane per frame");
		const unsigned odd = (unsigned)reinterpret_cast<uintptr_t>(p) & 2u;
		even_base = p - odd;
		odd_base = p - 2u + odd;         // frame 1 starts FB = 2 (mod 4) bytes on: in the other half
		even_shift = odd * 8u;
		odd_shift = 16u - odd * 8u;
	}

	__device__ __forceinline__ void load(Frame<CH> &f, int slot) const
	{
		constexpr int WORDS = Frame<CH>::WORDS;
		const unsigned *q = reinterpret_cast<const unsigned *>(((slot & 1) ? odd_base : even_base) + slot * FB);
		const unsigned shift = (slot & 1) ? odd_shift : even_shift;
		unsigned d[WORDS];
#pragma unroll
		for (int k = 0; k < WORDS; ++k)
			d[k] = q[k];
#pragma unroll
		for (int k = 0; k + 1 < WORDS; ++k)
			f.v[k] = (int)__builtin_amdgcn_alignbit(d[k + 1], d[k], shift);
		f.v[WORDS - 1] = (int)(d[WORDS - 1] >> shift);
	}
};

// NINT consecutive int32 to a destination that is only DWORD-aligned: 16-byte, then 8-byte, then 4-byte stores
// (stores_of_ints_dword_aligned(NINT) instructions).
typedef i32x4 i32x4_dword_aligned __attribute__((aligned(4)));
typedef i32x2 i32x2_dword_aligned __attribute__((aligned(4)));

// Non-temporal stores are for lanes whose bytes leave GAPLESS in ONE instruction whose lanes tile a contiguous stretch (4, 8, 12 or
// 16 bytes per lane).  (32 bytes per lane - two instructions, each writing every other 16-byte granule - gain 3-4 % from the hint
// on some boxes and lose 3-50 % on others: cfg 4's geometry sweep, the 8-channel chain trial; plain there too.)  A 24-byte frame leaves as 16 + 8 bytes at a
// stride of 24: each instruction writes part of every 16-byte granule, and marked non-temporal such partial lines may be written
// out before the other instruction has completed them - measured as a slow MODE some boxes / runs fall into and others do not
// (10 minutes of 6 channels at 44.1 -> 48 kHz: 186 us with plain stores on every box, 186 or 245 us with non-temporal ones;
// 12 channels 375 against 375 / 456 us; profiles/r02_nt_partial_lines.log), while 4 and 8 channels gain 2 % from the hint.
constexpr bool nt_suits_ints(int n, bool dword_aligned_only)
{
	return dword_aligned_only ? (n == 1 || n == 3) : n <= 4;
}

template <int NINT, int NT_ASKED>
__device__ __forceinline__ void store_ints_dword_aligned(int *dst, const int *v)
{
	constexpr int NT = NT_ASKED && nt_suits_ints(NINT, true);
	int c = 0;
#pragma unroll
	for (; c + 4 <= NINT; c += 4)
	{
		i32x4 q;
		q.x = v[c];
		q.y = v[c + 1];
		q.z = v[c + 2];
		q.w = v[c + 3];
		if constexpr (NT)
			__builtin_nontemporal_store(q, reinterpret_cast<i32x4_dword_aligned *>(dst + c));
		else
			*reinterpret_cast<i32x4_dword_aligned *>(dst + c) = q;
	}
	if constexpr (NINT % 4 >= 2)
	{
		i32x2 q;
		q.x = v[c];
		q.y = v[c + 1];
		if constexpr (NT)
			__builtin_nontemporal_store(q, reinterpret_cast<i32x2_dword_aligned *>(dst + c));
		else
			*reinterpret_cast<i32x2_dword_aligned *>(dst + c) = q;
		c += 2;
	}
	if constexpr (NINT % 2 == 1)
	{
		if constexpr (NT)
			__builtin_nontemporal_store(v[c], dst + c);
		else
			dst[c] = v[c];
	}
}

constexpr int stores_of_ints_dword_aligned(int n)
{
	return n / 4 + (n % 4) / 2 + n % 2;
}

// NINT consecutive int32 -> global memory, widest stores the size allows.  NT = 1 marks them non-temporal: the output
// is written once and never read by the kernel; on MI355X that is worth ~7 % of HBM throughput for the stereo stream
// (8-byte stores) and costs a few % with 16-byte stores, so it is part of the per-instance tuning.
template <int NINT, int NT_ASKED>
__device__ __forceinline__ void store_ints(int *dst, const int *v)
{
	constexpr int NT = NT_ASKED && nt_suits_ints(NINT, false);
	if constexpr (NINT % 4 == 0)
	{
#pragma unroll
		for (int c = 0; c < NINT; c += 4)
		{
			i32x4 q;
			q.x = v[c];
			q.y = v[c + 1];
			q.z = v[c + 2];
			q.w = v[c + 3];
			if constexpr (NT)
				__builtin_nontemporal_store(q, reinterpret_cast<i32x4 *>(dst + c));
			else
				*reinterpret_cast<i32x4 *>(dst + c) = q;
		}
	}
	else if constexpr (NINT % 2 == 0)
	{
#pragma unroll
		for (int c = 0; c < NINT; c += 2)
		{
			i32x2 q;
			q.x = v[c];
			q.y = v[c + 1];
			if constexpr (NT)
				__builtin_nontemporal_store(q, reinterpret_cast<i32x2 *>(dst + c));
			else
				*reinterpret_cast<i32x2 *>(dst + c) = q;
		}
	}
	else
	{
		// an odd count: the frames are only dword-aligned; 16- and 8-byte stores need no more than that on gfx950
		store_ints_dword_aligned<NINT, NT_ASKED>(dst, v);
	}
}

// The consumers of the reference clamp every sample to 16 bits in their output callback, to +-0x7FFF (note: -0x7FFF, not
// -0x8000; examples/low-level.c:69-80, examples/high-level.c:74-85).  Opt-in fused form of that callback: clamp and
// store int16, which also halves the write traffic.
__device__ __forceinline__ int clamp_s16(int v)
{
	return v > 0x7FFF ? 0x7FFF : (v < -0x7FFF ? -0x7FFF : v);
}

template <int NSHORT, int NT>
__device__ __forceinline__ void store_shorts(short *dst, const int *v)
{
	if constexpr (NSHORT % 2 == 0)
	{
		int packed[NSHORT / 2];
#pragma unroll
		for (int k = 0; k < NSHORT / 2; ++k)
			packed[k] = (clamp_s16(v[2 * k]) & 0xFFFF) | (clamp_s16(v[2 * k + 1]) << 16);
		store_ints<NSHORT / 2, NT>(reinterpret_cast<int *>(dst), packed);
	}
	else
	{
#pragma unroll
		for (int c = 0; c < NSHORT; ++c)
			dst[c] = (short)clamp_s16(v[c]);
	}
}

// The FEWEST store instructions a lane's `bytes` contiguous output bytes can leave as: 16 per instruction.  This, not the number
// of store statements in the source, is what a counted s_waitcnt vmcnt(N) behind a tile's stores may assume: hipcc merges
// neighbouring stores (the 8 + 4 bytes of a 3-channel frame leave as ONE global_store_dwordx3), and a count larger than the
// stores really outstanding no longer covers the LDS-DMA issued before them.  (Round 1 counted statements: with 3, 6 and 7
// channels the wait was too weak - never seen with k_poly's long tiles, caught by k_wave2's short ones.)  A count that is too
// small only waits for some of the tile's own stores as well.
constexpr int min_stores_of_bytes(int bytes)
{
	return (bytes + 15) / 16;
}

constexpr int stores_of_ints(int n)
{
	return n % 4 == 0 ? n / 4 : (n % 2 == 0 ? n / 2 : stores_of_ints_dword_aligned(n));
}

// ---------------------------------------------------------------------------------------------------------
// Row index of a fractional position (host mirror: cr_plan.c cr_poly_row_of)
// ---------------------------------------------------------------------------------------------------------
// `shift` receives the number of frames this phase's window starts after the tile's first window frame: affine rows are laid
// out from their own first tap (cr_plan.c, "SHIFTED windows"), and min_relative is computed here anyway.
template <int MODE>
__device__ __forceinline__ unsigned row_of(const crhip_poly_launch &a, unsigned frac, unsigned &shift)
{
	if constexpr (MODE == CRHIP_ROWMODE_UPSAMPLE)
	{
		shift = 0;
		return (65536u - frac) >> 6;
	}
	else
	{
		// min_relative / max_relative of clownresampler.h:993-994, kernel_start of :1001
		const unsigned mr = (frac + a.delta + 65535u) >> 16;
		const unsigned xr = (frac + a.skr) >> 16;
		const unsigned kstart = __umul24(a.step, (mr << 16) - frac) >> 16;
		shift = mr - a.first_mr;
		return (unsigned)((int)kstart + a.aff_a * (int)mr + a.aff_b * (int)xr + a.aff_c);
	}
}

// Everything one output frame reads from LDS, held in registers: its row (weights + reciprocal) and its window of
// input frames.  Splitting the frame into fetch_frame (LDS reads only) and compute_frame (VALU only) lets the kernel
// issue the reads of frame i+1 before the arithmetic of frame i: hipcc does not software-pipeline across the asm tap
// statements on its own, and with every wave of a workgroup released by the same barrier the waves otherwise alternate
// in lockstep between an LDS phase and a VALU phase.
template <int CH, int TT>
struct FrameData
{
	static constexpr int RS = (TT + 1 + 3) & ~3;
	int w[RS];
	Frame<CH> f[TT];
};

// PADDED tiles: run-time-slot instances with two lanes per frame where a lane's share does not start on a dword - 9, 10, 11, 13, 14
// and 15 channels (frames of 18 to 30 bytes).  Read where they lie, every tap costs a lane five dword reads
// and four funnel shifts; with long windows a frame is read by 15 to 33 taps of every output frame near it.  So k_poly repacks such
// a tile ONCE after its DMA has landed, LDS -> LDS: lane-share L = 2 * frame + half goes to L * 16 - frames of 32 bytes, every
// share one aligned ds_read_b128, its tail (the first samples of what follows) meeting the phantom channel or nothing.
template <int CH, int TT, int SPLIT, int PH>
constexpr bool padded_frames()
{
	return TT == 0 && SPLIT == 2 && CH >= 5 && (PH == 1 || CH % 2 == 1);   // (9, 10, 11, 13, 14, 15 channels)
}

// SPLIT > 1: a frame of CH * SPLIT channels is shared by SPLIT neighbouring lanes, each taking CH of them (`base` then
// already points at the lane's share of the first frame); FS is the distance between consecutive frames.
// PH: the frame has CH * SPLIT - 1 channels (k_poly's phantom channel): a lane's share then starts on any 2-byte boundary.
// FORM: 0 in every shipped instance; timing-only forms of k_poly (its ABL >> 4): bit 0 = every lane the same window, bit 1 = every lane row 0
template <int CH, int TT, int MODE, int SWZ, int SPLIT = 1, int PH = 0, int FORM = 0>
__device__ __forceinline__ void fetch_frame(const crhip_poly_launch &a, const int *rows, const unsigned char *base, unsigned rel, FrameData<CH, TT> &d)
{
	constexpr unsigned FB = (CH * SPLIT - PH) * 2;
	unsigned shift;
	const unsigned row = row_of<MODE>(a, rel & 0xFFFFu, shift);
	const unsigned phys = (FORM & 2) ? 0u : (SWZ ? ((row & ~15u) | ((__umul24(row >> 4, a.swizzle) + row) & 15u)) : row);
	const unsigned char *src = (FORM & 1) ? base : base + ((rel >> 16) + shift) * FB;
	const i32x4 *plane0 = reinterpret_cast<const i32x4 *>(rows) + phys;

#pragma unroll
	for (int q = 0; q < FrameData<CH, TT>::RS / 4; ++q)
	{
		const i32x4 v = plane0[q * a.plane_rows];
		d.w[4 * q] = v.x;
		d.w[4 * q + 1] = v.y;
		d.w[4 * q + 2] = v.z;
		d.w[4 * q + 3] = v.w;
	}
	if constexpr (CH == 1 && SPLIT == 1)
	{
		// mono: the window as packed pairs in f[0 .. (TT + 1) / 2) (see compute_frame)
		int pw[(TT + 1) / 2];
		load_mono_window<TT>(src, pw);
#pragma unroll
		for (int k = 0; k < (TT + 1) / 2; ++k)
			d.f[k].v[0] = pw[k];
	}
	else if constexpr (CH % 2 == 1 && SPLIT == 1)
	{
		const OddWindow<CH, (int)FB> window(src);
#pragma unroll
		for (int s = 0; s < TT; ++s)
			window.load(d.f[s], s);
	}
	else
	{
#pragma unroll
		for (int s = 0; s < TT; ++s)
		{
			if constexpr (PH)
				d.f[s].load_any(src + s * FB);
			else
				d.f[s].load(src + s * FB);
		}
	}
}

// The tap arithmetic as a chain of 64-bit multiply-adds (ASM mode 2; pure upsampling only, where the sign of a slot's weights is
// a compile-time property - NEGMASK in ASM >> 8, checked by the host against the plan's rows), in k_wave2's mov-armed form:
//     X = 2 * sample                    one SDWA shift straight from the packed frame (left: low word, right: high word)
//     P = (accumulator : X)             high dword: the running sum; low dword: X itself - in [2^32 - 65536, 2^32) where the sample
//                                       is negative, in [0, 65536) where it is not, which is all the carry needs (a v_mov_b32, the
//                                       one arming instruction that is nearly free beside the multiply-add: chainbench forms 12-19)
//     P = v_mad_i64_i32(X, W, P)        W = |weight| << 15 as the kernel staged it: X * W = sample * |weight| * 65536, the integer part
//                                       of sample * |weight| / 65536 lands in the high dword, truncated toward zero
// (clownresampler.h:1020 via :625).  Slots with negative weights run on a second chain whose sum is subtracted at the end
// (truncation toward zero is odd-symmetric) - which also keeps consecutive multiply-adds independent.  |weight| << 15 needs
// |weight| < 65536: the slots in mad_safemask<TT>() - the two around the kernel's centre, whose weights reach 65536 in one row each -
// are staged as plain |weight| and take X << 15 = sample << 16 with the sign-shift arm.
typedef short s16x2 __attribute__((ext_vector_type(2)));

template <int TT>
constexpr unsigned mad_safemask()
{
	return TT == 5 ? 0xCu : (TT == 15 ? 0x180u : 0u);
}

// what a kernel does to weight `e` of slot `slot` while it stages the rows for this arithmetic
template <int TT, unsigned NEGMASK>
__device__ __forceinline__ int mad_staged_weight(int e, int slot)
{
	if (slot < TT && ((NEGMASK >> slot) & 1u))
		e = -e;
	if (slot < TT && !((mad_safemask<TT>() >> slot) & 1u))
		e = (int)((unsigned)e << 15);
	return e;
}

// p = x * weight + p, 64 bits, one instruction.  Inline asm: left to itself hipcc re-associates the low dword out of the addend and
// adds it with a separate 64-bit add.
__device__ __forceinline__ void mad64(i32x2 &p, int x, int weight)
{
	long long carry;
	asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(p), "=&s"(carry) : "v"(x), "v"(weight));
}

template <bool SAFE, bool FIRST>
__device__ __forceinline__ void mad64_tap_pair(i32x2 &p_lo, i32x2 &p_hi, int frame, int weight)
{
	if (FIRST)
	{
		p_lo.y = 0;
		p_hi.y = 0;
	}
	if constexpr (SAFE)
	{
		const int xl = (int)((unsigned)frame << 16), xr = (int)((unsigned)frame & 0xFFFF0000u);
		p_lo.x = xl >> 31;
		p_hi.x = xr >> 31;
		mad64(p_lo, xl, weight);
		mad64(p_hi, xr, weight);
	}
	else
	{
		int xl, xr;   // 2 * sample, sign-extended
		asm("v_lshlrev_b32_sdwa %0, %2, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(xl) : "v"(frame), "v"(1));
		asm("v_lshlrev_b32_sdwa %0, %2, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(xr) : "v"(frame), "v"(1));
		p_lo.x = xl;
		p_hi.x = xr;
		mad64(p_lo, xl, weight);
		mad64(p_hi, xr, weight);
	}
}

// The same chain for ANY rows (ASM mode 3; downsampling, where a slot's weight changes sign with the phase): the weight as it is,
// X = sample << 16 (a plain shift for the left channel, a plain mask for the right one) and the low dword armed with
// sext(top byte of the sample) ^ sext(top byte of the weight) - k_wave2's one SDWA instruction, reading the sample's top byte
// straight out of the packed frame (byte 1 / byte 3).  3 instructions per tap and channel against the 4 of the SDWA form; one
// chain per channel.
template <bool FIRST>
__device__ __forceinline__ void mad64_tap_pair_signed(i32x2 &p_lo, i32x2 &p_hi, int frame, int weight)
{
	if (FIRST)
	{
		p_lo.y = 0;
		p_hi.y = 0;
	}
	const int xl = (int)((unsigned)frame << 16), xr = (int)((unsigned)frame & 0xFFFF0000u);
	int al, ar;
	asm("v_xor_b32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_3" : "=v"(al) : "v"(frame), "v"(weight));
	asm("v_xor_b32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(ar) : "v"(frame), "v"(weight));
	p_lo.x = al;
	p_hi.x = ar;
	mad64(p_lo, xl, weight);
	mad64(p_hi, xr, weight);
}

// ... one channel alone (the last one of an odd frame: the low half of its dword)
__device__ __forceinline__ void mad64_tap_word0_signed(i32x2 &p, int packed, int weight)
{
	const int x = (int)((unsigned)packed << 16);
	int arm;
	asm("v_xor_b32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_3" : "=v"(arm) : "v"(packed), "v"(weight));
	p.x = arm;
	mad64(p, x, weight);
}

// ... mono, where a packed dword holds two FRAMES: word WORD of `packed`
template <int WORD>
__device__ __forceinline__ void mad64_tap_mono_signed(i32x2 &p, int packed, int weight)
{
	const int x = WORD ? (int)((unsigned)packed & 0xFFFF0000u) : (int)((unsigned)packed << 16);
	int arm;
	if constexpr (WORD)
		asm("v_xor_b32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(arm) : "v"(packed), "v"(weight));
	else
		asm("v_xor_b32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_3" : "=v"(arm) : "v"(packed), "v"(weight));
	p.x = arm;
	mad64(p, x, weight);
}

// a whole frame's tap on one chain per channel (the chains' high dwords must have been zeroed before the first tap)
template <int CH>
__device__ __forceinline__ void frame_tap_signed(i32x2 (&p)[CH], const Frame<CH> &f, int weight)
{
#pragma unroll
	for (int k = 0; k < CH / 2; ++k)
		mad64_tap_pair_signed<false>(p[2 * k], p[2 * k + 1], f.v[k], weight);
	if constexpr (CH % 2 == 1)
		mad64_tap_word0_signed(p[CH - 1], f.v[CH / 2], weight);
}

// ... and for mono, where a packed dword holds two FRAMES (two slots, each with its own weight and slot class): word WORD of `packed`
template <bool SAFE, bool FIRST, int WORD>
__device__ __forceinline__ void mad64_tap_mono(i32x2 &p, int packed, int weight)
{
	if (FIRST)
		p.y = 0;
	if constexpr (SAFE)
	{
		const int x = WORD ? (int)((unsigned)packed & 0xFFFF0000u) : (int)((unsigned)packed << 16);
		p.x = x >> 31;
		mad64(p, x, weight);
	}
	else
	{
		int x;   // 2 * sample, sign-extended
		if constexpr (WORD)
			asm("v_lshlrev_b32_sdwa %0, %2, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(x) : "v"(packed), "v"(1));
		else
			asm("v_lshlrev_b32_sdwa %0, %2, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(x) : "v"(packed), "v"(1));
		p.x = x;
		mad64(p, x, weight);
	}
}

template <int CH, int TT, int NORM, int ASM>
__device__ __forceinline__ void compute_frame(const FrameData<CH, TT> &d, int *out)
{
	if constexpr ((ASM & 0xFF) == 2 && CH == 1)
	{
		constexpr unsigned NEGMASK = (unsigned)ASM >> 8;
		constexpr unsigned SAFEMASK = mad_safemask<TT>();
		static_assert(SAFEMASK != 0 && (SAFEMASK & NEGMASK) == 0 && NEGMASK != 0, "slot classes of the instance");
		constexpr int FIRST_POS = __builtin_ctz(~NEGMASK), FIRST_NEG = __builtin_ctz(NEGMASK);
		static_assert(!((SAFEMASK >> FIRST_POS) & 1u), "a chain's first tap is an ordinary slot");
		i32x2 p, p2;   // the chain of the positive slots and the chain of the negative ones
#pragma unroll
		for (int s = 0; s < TT; ++s)
		{
			const int packed = d.f[s / 2].v[0];   // fetch_frame left the window packed, two frames per dword
			if ((NEGMASK >> s) & 1u)
			{
				if (s == FIRST_NEG)
					(s & 1) ? mad64_tap_mono<false, true, 1>(p2, packed, d.w[s]) : mad64_tap_mono<false, true, 0>(p2, packed, d.w[s]);
				else
					(s & 1) ? mad64_tap_mono<false, false, 1>(p2, packed, d.w[s]) : mad64_tap_mono<false, false, 0>(p2, packed, d.w[s]);
			}
			else if (s == FIRST_POS)
				(s & 1) ? mad64_tap_mono<false, true, 1>(p, packed, d.w[s]) : mad64_tap_mono<false, true, 0>(p, packed, d.w[s]);
			else if ((SAFEMASK >> s) & 1u)
				(s & 1) ? mad64_tap_mono<true, false, 1>(p, packed, d.w[s]) : mad64_tap_mono<true, false, 0>(p, packed, d.w[s]);
			else
				(s & 1) ? mad64_tap_mono<false, false, 1>(p, packed, d.w[s]) : mad64_tap_mono<false, false, 0>(p, packed, d.w[s]);
		}
		out[0] = normalise<NORM>(p.y - p2.y, d.w[TT]);
		return;
	}
	else if constexpr ((ASM & 0xFF) == 3 && CH == 1)
	{
		// mono: fetch_frame left the window packed, two frames per dword; two chains, taps alternating between them
		i32x2 p, p2;
		p.y = 0;
		p2.y = 0;
#pragma unroll
		for (int s = 0; s < TT; ++s)
		{
			const int packed = d.f[s / 2].v[0];
			if (s & 1)
				mad64_tap_mono_signed<1>(p2, packed, d.w[s]);
			else
				mad64_tap_mono_signed<0>(p, packed, d.w[s]);
		}
		out[0] = normalise<NORM>(p.y + p2.y, d.w[TT]);
		return;
	}
	else if constexpr ((ASM & 0xFF) == 3)
	{
		i32x2 p[CH];
#pragma unroll
		for (int c = 0; c < CH; ++c)
			p[c].y = 0;
#pragma unroll
		for (int s = 0; s < TT; ++s)
			frame_tap_signed<CH>(p, d.f[s], d.w[s]);
#pragma unroll
		for (int c = 0; c < CH; ++c)
			out[c] = normalise<NORM>(p[c].y, d.w[TT]);
		return;
	}
	else if constexpr ((ASM & 0xFF) == 2)
	{
		static_assert(CH % 2 == 0, "the 64-bit chain works on packed pairs of channels (or on mono: above)");
		constexpr unsigned NEGMASK = (unsigned)ASM >> 8;
		constexpr unsigned SAFEMASK = mad_safemask<TT>();
		static_assert(SAFEMASK != 0 && (SAFEMASK & NEGMASK) == 0 && NEGMASK != 0 && (~NEGMASK & ((1u << TT) - 1u)) != 0, "slot classes of the instance");
		constexpr int FIRST_POS = __builtin_ctz(~NEGMASK), FIRST_NEG = __builtin_ctz(NEGMASK);
		static_assert(!((SAFEMASK >> FIRST_POS) & 1u), "a chain's first tap is an ordinary slot");
		i32x2 p[CH], p2[CH];   // the chain of the positive slots and the chain of the negative ones
#pragma unroll
		for (int s = 0; s < TT; ++s)
		{
#pragma unroll
			for (int k = 0; k < CH / 2; ++k)
			{
				if ((NEGMASK >> s) & 1u)
				{
					if (s == FIRST_NEG)
						mad64_tap_pair<false, true>(p2[2 * k], p2[2 * k + 1], d.f[s].v[k], d.w[s]);
					else
						mad64_tap_pair<false, false>(p2[2 * k], p2[2 * k + 1], d.f[s].v[k], d.w[s]);
				}
				else if (s == FIRST_POS)
					mad64_tap_pair<false, true>(p[2 * k], p[2 * k + 1], d.f[s].v[k], d.w[s]);
				else if ((SAFEMASK >> s) & 1u)
					mad64_tap_pair<true, false>(p[2 * k], p[2 * k + 1], d.f[s].v[k], d.w[s]);
				else
					mad64_tap_pair<false, false>(p[2 * k], p[2 * k + 1], d.f[s].v[k], d.w[s]);
			}
		}
#pragma unroll
		for (int c = 0; c < CH; ++c)
		{
			const int acc = p[c].y - p2[c].y;
			out[c] = normalise<NORM>(acc, d.w[TT]);
		}
		return;
	}
	if constexpr (CH == 1)
	{
		// mono: fetch_frame left the window packed, two frames per dword
		int pw[(TT + 1) / 2];
#pragma unroll
		for (int k = 0; k < (TT + 1) / 2; ++k)
			pw[k] = d.f[k].v[0];
		int a0 = 0, a1 = 0;
#pragma unroll
		for (int s = 0; s < TT; ++s)
		{
			if (s == 0)
				mono_tap<ASM, true>(a0, pw, s, d.w[s]);
			else if (s == 1)
				mono_tap<ASM, true>(a1, pw, s, d.w[s]);
			else if (s & 1)
				mono_tap<ASM, false>(a1, pw, s, d.w[s]);
			else
				mono_tap<ASM, false>(a0, pw, s, d.w[s]);
		}
		out[0] = normalise<NORM>(a0 + a1, d.w[TT]);
		return;
	}
	// two accumulator sets, taps alternating between them: consecutive tap statements are independent (no asm boundary
	// pad, more overlap); integer addition is associative, so the sum is the same
	int acc[CH], acc2[CH];
#pragma unroll
	for (int s = 0; s < TT; ++s)
	{
		if (s == 0)
			d.f[s].template mac_first<ASM>(acc, d.w[s]);
		else if (s == 1)
			d.f[s].template mac_first<ASM>(acc2, d.w[s]);
		else if (s & 1)
			d.f[s].template mac<ASM>(acc2, d.w[s]);
		else
			d.f[s].template mac<ASM>(acc, d.w[s]);
	}
	if constexpr (TT > 1)
	{
#pragma unroll
		for (int c = 0; c < CH; ++c)
			acc[c] += acc2[c];
	}
#pragma unroll
	for (int c = 0; c < CH; ++c)
		out[c] = normalise<NORM>(acc[c], d.w[TT]);
}

// One output frame: CH normalised int32 into out[0..CH).
//   rel   16.16 position relative to the tile's first integer position
//   base  LDS address of the tile's first window frame (tile + shift)
template <int CH, int TT, int MODE, int NORM, int ASM, int SWZ, int SPLIT = 1, int PH = 0, int PADT = 0>
__device__ __forceinline__ void one_frame(const crhip_poly_launch &a, const int *rows, const unsigned char *base, unsigned rel, int *out)
{
	static_assert(PH == 0 || SPLIT == 2, "the phantom channel exists for instances with two lanes per frame");
	if constexpr ((ASM & 0xFF) == 2 || ((ASM & 0xFF) == 3 && TT > 0))
	{
		// (the rows are staged for the 64-bit chain: every path of such an instance goes through it - and so does the any-sign chain
		// of a specialised instance)
		FrameData<CH, TT> d;
		fetch_frame<CH, TT, MODE, SWZ, SPLIT, PH>(a, rows, base, rel, d);
		compute_frame<CH, TT, NORM, ASM>(d, out);
		return;
	}
	constexpr bool PAD = PADT != 0;   // (k_poly's padded tiles: frames of 32 bytes, a lane's share 16)
	constexpr unsigned FB = PAD ? 32u : (CH * SPLIT - PH) * 2;
	constexpr int RS_CT = (TT + 1 + 3) & ~3;

	const unsigned frac = rel & 0xFFFFu;
	unsigned shift;
	const unsigned row = row_of<MODE>(a, frac, shift);
	// LDS image of the rows: PLANAR (plane q holds int32 [4q, 4q+4) of every row, 16 bytes per row) and SWIZZLED
	// within each block of 16 rows by a host-chosen multiple of the block number, so that the 16 lanes a
	// ds_read_b128 services together fall on 16 different 16-byte bank slots instead of the 5-8 the plain layout
	// gives for a fixed increment (cr_plan.c cr_poly_pick_swizzle).
	const unsigned phys = SWZ ? ((row & ~15u) | ((__umul24(row >> 4, a.swizzle) + row) & 15u)) : row;
	const unsigned char *src = base + ((rel >> 16) + shift) * FB;
	const i32x4 *plane0 = reinterpret_cast<const i32x4 *>(rows) + phys;

	int acc[CH];
#pragma unroll
	for (int c = 0; c < CH; ++c)
		acc[c] = 0;
	// ASM mode 3 (run-time slot count): one 64-bit chain per channel instead of the SDWA taps - 3 instructions per tap and channel, not 4
	constexpr bool CHAIN3 = (ASM & 0xFF) == 3 && !(CH == 1 && SPLIT == 1);
	i32x2 chain[CHAIN3 ? CH : 1];
	i32x2 mono_chain;   // (mono with a run-time slot count: its packed-window taps on one chain)
	mono_chain.y = 0;
	if constexpr (CHAIN3)
	{
#pragma unroll
		for (int c = 0; c < CH; ++c)
			chain[c].y = 0;
	}

	int reciprocal = 0;

	if constexpr (TT > 0)
	{
		int w[RS_CT];
#pragma unroll
		for (int q = 0; q < RS_CT / 4; ++q)
		{
			const i32x4 v = plane0[q * a.plane_rows];
			w[4 * q] = v.x;
			w[4 * q + 1] = v.y;
			w[4 * q + 2] = v.z;
			w[4 * q + 3] = v.w;
		}
		// two accumulator sets, taps alternating between them: consecutive tap statements are then independent (no asm
		// boundary pad, more overlap); integer addition is associative, so the sum is the same
		int acc2[CH];
		int pw[CH == 1 && SPLIT == 1 ? (TT + 1) / 2 : 1];
		if constexpr (CH == 1 && SPLIT == 1)
			load_mono_window<TT>(src, pw);
#pragma unroll
		for (int s = 0; s < TT; ++s)
		{
			if constexpr (CH == 1 && SPLIT == 1)
			{
				if (s == 0)
					mono_tap<ASM, true>(acc[0], pw, s, w[s]);
				else if (s == 1)
					mono_tap<ASM, true>(acc2[0], pw, s, w[s]);
				else
					mono_tap<ASM, false>((s & 1) ? acc2[0] : acc[0], pw, s, w[s]);
				continue;
			}
			Frame<CH> f;
			if constexpr (CH % 2 == 1 && SPLIT == 1)
				OddWindow<CH, (int)FB>(src).load(f, s);
			else if constexpr (PH)
				f.load_any(src + s * FB);
			else
				f.load(src + s * FB);
			if (s == 0)
				f.template mac_first<ASM>(acc, w[s]);
			else if (s == 1)
				f.template mac_first<ASM>(acc2, w[s]);
			else if (s & 1)
				f.template mac<ASM>(acc2, w[s]);
			else
				f.template mac<ASM>(acc, w[s]);
		}
		if constexpr (TT > 1)
		{
#pragma unroll
			for (int c = 0; c < CH; ++c)
				acc[c] += acc2[c];
		}
		reciprocal = w[TT];
	}
	else
	{
		// Run-time slot count.  The device image of the rows is laid out for this loop (cr_plan.c, SPLIT layout):
		// ceil(slots / 4) planes of weights, zero-padded, then one plane that holds only the reciprocal.  Four taps
		// per trip - one ds_read_b128 of weights, four frame reads, four independent multiply-accumulates - and, where it
		// measured faster (TAIL below), a last SHORTER trip for slots % 4 taps (a wave-uniform switch): 5 slots - every pure
		// upsampling with three lobes - are then 5 taps of arithmetic, not 8.
		const unsigned weight_planes = a.row_stride / 4u - 1u;
		auto trip = [&](unsigned q, auto count_tag) {
			constexpr int N = decltype(count_tag)::value;   // taps of this trip, 1..4
			const i32x4 v = plane0[q * a.plane_rows];
			const int wv[4] = {v.x, v.y, v.z, v.w};
			if constexpr (CH == 1 && SPLIT == 1)
			{
				// mono: the frames of this trip as packed dwords (the parity of the window's start is the same in every trip:
				// four frames are 8 bytes)
				int pw[2];
				load_mono_window<N>(src + 4u * q * FB, pw);
#pragma unroll
				for (int k = 0; k < N; ++k)
				{
					if constexpr ((ASM & 0xFF) == 3)
						(k & 1) ? mad64_tap_mono_signed<1>(mono_chain, pw[k / 2], wv[k]) : mad64_tap_mono_signed<0>(mono_chain, pw[k / 2], wv[k]);
					else
						mono_tap<ASM, false>(acc[0], pw, k, wv[k]);
				}
			}
			else
			{
				Frame<CH> f[N];
				if constexpr (CH % 2 == 1 && SPLIT == 1)
				{
					// four frames of an odd channel count are 4 * FB = 0 (mod 8) bytes: the window of every trip starts alike
					const OddWindow<CH, (int)FB> window(src + 4u * q * FB);
#pragma unroll
					for (int k = 0; k < N; ++k)
						window.load(f[k], k);
				}
				else
				{
#pragma unroll
					for (int k = 0; k < N; ++k)
					{
						if constexpr (PAD)
							f[k].load_padded(src + (4u * q + (unsigned)k) * FB);
						else if constexpr (PH)
							f[k].load_any(src + (4u * q + (unsigned)k) * FB);
						else
							f[k].load(src + (4u * q + (unsigned)k) * FB);
					}
				}
#pragma unroll
				for (int k = 0; k < N; ++k)
				{
					if constexpr (CHAIN3)
						frame_tap_signed<CH>(chain, f[k], wv[k]);
					else
						f[k].template mac<ASM>(acc, wv[k]);
				}
			}
		};
		// What to do with the slots % 4 taps of the last plane was MEASURED per lane shape (profiles/r01_channel_table.log;
		// 44.1 -> 48 kHz, 5 slots): the shorter last trip gains 10-30 % up to 6 channels per lane; 7-8 channels per lane and
		// most phantom shapes LOSE 10-30 % with it (three more unrolled trip bodies in every straight-line frame of a tile),
		// so there the zero-padded full trip stays (a padded slot has weight 0 and contributes exactly 0 whatever the LDS
		// read returns).
		constexpr bool SHORT_TAIL = (CH <= 6 && !PH) || (CH == 5 && PH);
		const unsigned full_trips = SHORT_TAIL ? a.slots / 4u : weight_planes;
		for (unsigned q = 0; q < full_trips; ++q)
			trip(q, std::integral_constant<int, 4>());
		if constexpr (SHORT_TAIL)
		{
			switch (a.slots & 3u)
			{
				case 1: trip(full_trips, std::integral_constant<int, 1>()); break;
				case 2: trip(full_trips, std::integral_constant<int, 2>()); break;
				case 3: trip(full_trips, std::integral_constant<int, 3>()); break;
				default: break;
			}
		}
		reciprocal = reinterpret_cast<const int *>(plane0 + weight_planes * a.plane_rows)[0];
		if constexpr (CHAIN3)
		{
#pragma unroll
			for (int c = 0; c < CH; ++c)
				acc[c] = chain[c].y;
		}
		else if constexpr ((ASM & 0xFF) == 3)
			acc[0] = mono_chain.y;
	}

#pragma unroll
	for (int c = 0; c < CH; ++c)
		out[c] = normalise<NORM>(acc[c], reciprocal);
}

// ---------------------------------------------------------------------------------------------------------
// Tickets.  draw_ticket returns the counter's value before the increment, once per WAVE (a scalar instruction: every lane of
// the calling wave sees the same value; call it from wave-uniform control flow only).  It is s_atomic_add: the value comes
// back through lgkmcnt into an SGPR and the wait for it, here, is this wave's own round trip to the L2 and nothing else.  The
// obvious form - lane 0 does a vector atomic - returns through vmcnt into a VGPR, and hipcc waits for that register with
// vmcnt(0) long before it is used (it rewrites a single-lane atomic into the wave-aggregated form, whose readfirstlane needs
// the value on the spot; and with that turned off it still flushes at the head of any loop that reads the register): the
// wave's outstanding stores and LDS-DMA are drained once per ticket.  tools/microbench/satomic.hip checks that scalar atomics
// exist on gfx950 and hand out dense, unique values next to vector atomics on the same word.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned draw_ticket(unsigned *counter)
{
	unsigned got = 1u;
	asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(got) : "s"(counter) : "memory");
	return got;
}

// The same in two halves, for a caller that has a wait of its own to put in between (the round trip then runs under it).  The
// SGPR is in flight between the two: NOTHING but the caller's own inline-assembly waits may stand between them - hipcc knows
// nothing of the pending write and would read the register early if it had a reason to touch it.
__device__ __forceinline__ unsigned draw_ticket_begin(unsigned *counter)
{
	unsigned got = 1u;
	asm volatile("s_atomic_add %0, %1, 0x0 glc" : "+s"(got) : "s"(counter) : "memory");
	return got;
}
__device__ __forceinline__ unsigned draw_ticket_end(unsigned got)
{
	asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(got) : : "memory");
	return got;
}

// ---------------------------------------------------------------------------------------------------------
// SGPR budget.  Two 1024-thread workgroups per CU are 8 waves per SIMD, and on gfx950 a SIMD admits
// min(8, 800 / (ceil(sgpr_count / 16) * 16 + 16)) waves (MI355X_MICROARCH.md, "Residency"): 8 only up to .sgpr_count 80,
// 7 up to 96 - while the compiler's "Occupancy: 8" and hipOccupancyMaxActiveBlocksPerMultiprocessor still say 8 / two
// workgroups.  Left alone hipcc gave most instances 82-106 SGPRs (wave-uniform 64-bit addresses, tile bookkeeping), so the
// second workgroup of a CU only started when the first had finished - seen in the start ticks of the stamped diagnostic
// instance.  80 includes VCC, FLAT_SCRATCH and XNACK_MASK; what does not fit is spilled to VGPR lanes, of which there are
// plenty (38-63 of 64 used).
// ---------------------------------------------------------------------------------------------------------
#define CRHIP_SGPR_BUDGET 80

// ---------------------------------------------------------------------------------------------------------
// One architecture.  Several kernels drop stores that would land behind the caller's buffer by the RANGE CHECK of a raw buffer descriptor
// with the per-store offset in the instruction's SCALAR offset operand (k_poly's and k_wave2's dual-mono stores, k_up2's copy-out,
// k_seg's segments): on gfx9 / gfx950 the check is "offset >= num_records - soffset", the scalar offset included.  Other targets
// leave it out of the check (LLVM's AMDGPU usage notes) and would write out of bounds - so this code is refused anywhere else
// (ADVICE r4), and the guard-region tests (tests/test_gpu_parity.py: dual mono, test_segment_kernel_bit_exact) stay a mandatory gate.
// ---------------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libclownresampler_amd's kernels are written for gfx950 (MI355X) only: see the note on buffer range checks above"
#endif

} // namespace

#endif // CR_DEVICE_HPP
