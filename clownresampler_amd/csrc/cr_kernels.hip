// cr_kernels.hip - the windowed-sinc hot path on CDNA4 (gfx950) + the thin C-ABI launch shim (crhip.h).
//
// What runs here is the body of the reference's per-output-frame loop: ClownResampler_LowestLevel_Resample
// (reference clownresampler.h:986-1035) evaluated for every output frame that
// ClownResampler_LowLevel_Resample (clownresampler.h:1058-1092) would walk to.  Output frames are independent:
// frame j sits at the 16.16 position pos0 + j * increment (closed form of clownresampler.h:1076-1078).
//
// Three kernels:
//
//  k_poly     The fast path.  The host (cr_plan.c) re-indexes the caller's Lanczos table into POLYPHASE ROWS:
//             every fractional position maps to one row holding the `slots` weights that position uses
//             (zero-padded to a common window) followed by the exact 17.15 reciprocal of their sum
//             (clownresampler.h:1025), so the device does neither the strided table walk nor the integer divide.
//             A persistent workgroup stages the rows in LDS once, then takes tiles of output frames in stream order
//             (the first by its own number, the rest as atomic tickets - or plain round-robin where that measured
//             better): the input PCM window of the NEXT tile is fetched by LDS-DMA (16-byte `buffer_load ... lds`,
//             bounds-checked by a per-tile buffer descriptor) into the other half of a double-buffered LDS tile while
//             the current tile is computed; each lane produces whole output frames from LDS (weights: ds_read_b128
//             of its row; samples: one LDS read per tap covering all channels of the frame; the reads of frame i+1
//             are issued before the arithmetic of frame i) and stores them non-temporally.  Per tap and channel the
//             arithmetic is v_mul_i32_i24 + truncate-toward-zero /65536 + add, in that order: the reference
//             truncates every product BEFORE accumulating (clownresampler.h:1020), which is what rules out
//             dot-product instructions and MFMA.  All of it is 32-bit: the host only selects this kernel when it has
//             proved the bounds (-65536 < weight <= 65536, |acc| < 2^23, |acc * reciprocal| < 2^31 or 2^32).
//
//  k_wave     The same rows and arithmetic without any workgroup barrier after the staging: every wave streams
//             wave-tiles of 256 frames through its own double-buffered 1 KiB of LDS and draws its own tickets.
//             Faster than k_poly where the arithmetic dominates (8-lobe stereo), slower where memory does.
//
//  k_generic  The reference arithmetic restated with 64-bit integers, one lane per output frame, weights read
//             from the original table in global memory.  It takes every configuration the reference accepts and
//             the accumulate-into semantics of ClownResampler_LowestLevel_Resample; it is what runs when k_poly's
//             preconditions do not hold, and it doubles as an independent second implementation in the tests.
//
// gfx950 only.  No CUDA/HIP dual paths.

#include <hip/hip_runtime.h>

#include <stdint.h>
#include <string.h>

#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#include "crhip.h"

namespace
{

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

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

template <int NINT, int NT>
__device__ __forceinline__ void store_ints_dword_aligned(int *dst, const int *v)
{
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
template <int NINT, int NT>
__device__ __forceinline__ void store_ints(int *dst, const int *v)
{
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
		store_ints_dword_aligned<NINT, NT>(dst, v);
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

// SPLIT > 1: a frame of CH * SPLIT channels is shared by SPLIT neighbouring lanes, each taking CH of them (`base` then
// already points at the lane's share of the first frame); FS is the distance between consecutive frames.
template <int CH, int TT, int MODE, int SWZ, int SPLIT = 1>
__device__ __forceinline__ void fetch_frame(const crhip_poly_launch &a, const int *rows, const unsigned char *base, unsigned rel, FrameData<CH, TT> &d)
{
	constexpr unsigned FB = CH * 2 * SPLIT;
	unsigned shift;
	const unsigned row = row_of<MODE>(a, rel & 0xFFFFu, shift);
	const unsigned phys = SWZ ? ((row & ~15u) | ((__umul24(row >> 4, a.swizzle) + row) & 15u)) : row;
	const unsigned char *src = base + ((rel >> 16) + shift) * FB;
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
			d.f[s].load(src + s * FB);
	}
}

// The tap arithmetic as a chain of full-rate 64-bit multiply-adds (ASM mode 2; pure upsampling only, where the sign of a
// slot's weights is a compile-time property - NEGMASK in ASM >> 8, checked by the host against the plan's rows):
//     P = (accumulator : bias)          high dword: the running sum; low dword: 0xFFFF0000 where sample * weight < 0, else 0
//     P = v_mad_i64_i32(sample << 16, weight, P)
// The product lands as (sample * weight) << 16, so with the bias beside it the carry into the high dword is exactly the
// reference's (sample * weight) / 65536 with C truncation (clownresampler.h:1020 via :625) added to the running sum; what
// is left in the low dword is overwritten by the next tap's bias.  Per packed pair of channels and tap: one v_pk_ashrrev_i16
// (the sign masks of both samples; a zero sample may carry the bias too: (0 + 0xFFFF) >> 16 == 0), two shifts/masks for the
// samples, two for the biases, two multiply-adds: ~22 cycles per wave against ~30 for the SDWA form.
typedef short s16x2 __attribute__((ext_vector_type(2)));

// p = (sample << 16) * weight + p, 64 bits, one instruction.  Inline asm: left to itself hipcc re-associates the bias out of the
// addend and adds it with a separate 64-bit add.
__device__ __forceinline__ void mad64(i32x2 &p, int sample_shifted, int weight)
{
	long long carry;
	asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(p), "=&s"(carry) : "v"(sample_shifted), "v"(weight));
}

template <bool NEGATIVE_SLOT, bool FIRST>
__device__ __forceinline__ void mad64_tap_pair(i32x2 &p_lo, i32x2 &p_hi, int frame, int weight)
{
	const int seen = NEGATIVE_SLOT ? ~frame : frame;   // negative slot: the product is negative where the sample is positive
	const s16x2 masks = __builtin_bit_cast(s16x2, seen) >> (short)15;
	const unsigned pm = __builtin_bit_cast(unsigned, masks);
	// the bias goes straight into the low dword of the accumulator pair (x); the high dword (y) is the running sum
	p_lo.x = (int)(pm << 16);
	p_hi.x = (int)(pm & 0xFFFF0000u);
	if (FIRST)
	{
		p_lo.y = 0;
		p_hi.y = 0;
	}
	mad64(p_lo, (int)((unsigned)frame << 16), weight);
	mad64(p_hi, (int)((unsigned)frame & 0xFFFF0000u), weight);
}

template <unsigned NEGMASK, bool FIRST>
__device__ __forceinline__ void mad64_tap_pair_dispatch(int slot, i32x2 &p_lo, i32x2 &p_hi, int frame, int weight)
{
	if ((NEGMASK >> slot) & 1u)
		mad64_tap_pair<true, FIRST>(p_lo, p_hi, frame, weight);
	else
		mad64_tap_pair<false, FIRST>(p_lo, p_hi, frame, weight);
}

template <int CH, int TT, int NORM, int ASM>
__device__ __forceinline__ void compute_frame(const FrameData<CH, TT> &d, int *out)
{
	if constexpr ((ASM & 0xFF) == 2)
	{
		static_assert(CH % 2 == 0, "the 64-bit chain works on packed pairs of channels");
		constexpr unsigned NEGMASK = (unsigned)ASM >> 8;
		i32x2 p[CH], p2[CH];   // two chains, taps alternating: consecutive multiply-adds are independent
#pragma unroll
		for (int s = 0; s < TT; ++s)
		{
#pragma unroll
			for (int k = 0; k < CH / 2; ++k)
			{
				if (s == 0)
					mad64_tap_pair_dispatch<NEGMASK, true>(s, p[2 * k], p[2 * k + 1], d.f[s].v[k], d.w[s]);
				else if (s == 1)
					mad64_tap_pair_dispatch<NEGMASK, true>(s, p2[2 * k], p2[2 * k + 1], d.f[s].v[k], d.w[s]);
				else if (s & 1)
					mad64_tap_pair_dispatch<NEGMASK, false>(s, p2[2 * k], p2[2 * k + 1], d.f[s].v[k], d.w[s]);
				else
					mad64_tap_pair_dispatch<NEGMASK, false>(s, p[2 * k], p[2 * k + 1], d.f[s].v[k], d.w[s]);
			}
		}
#pragma unroll
		for (int c = 0; c < CH; ++c)
		{
			const int acc = p[c].y + (TT > 1 ? p2[c].y : 0);
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
template <int CH, int TT, int MODE, int NORM, int ASM, int SWZ, int SPLIT = 1, int PH = 0>
__device__ __forceinline__ void one_frame(const crhip_poly_launch &a, const int *rows, const unsigned char *base, unsigned rel, int *out)
{
	static_assert(PH == 0 || (TT == 0 && SPLIT == 2), "the phantom channel exists for run-time-slot instances with two lanes per frame");
	constexpr unsigned FB = (CH * SPLIT - PH) * 2;
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
					mono_tap<ASM, false>(acc[0], pw, k, wv[k]);
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
						if constexpr (PH)
							f[k].load_any(src + (4u * q + (unsigned)k) * FB);
						else
							f[k].load(src + (4u * q + (unsigned)k) * FB);
					}
				}
#pragma unroll
				for (int k = 0; k < N; ++k)
					f[k].template mac<ASM>(acc, wv[k]);
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
	}

#pragma unroll
	for (int c = 0; c < CH; ++c)
		out[c] = normalise<NORM>(acc[c], reciprocal);
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
//           launch flag of tools/): 1 = no output stores, 2 = no input DMA, 3 = neither, 4 = DMA + stores but no arithmetic
// OUT16     1 = clamp to +-0x7FFF and store int16 (opt-in extension), 0 = the reference's unclamped int32
// NT        1 = non-temporal output stores
// SPLIT     lanes per frame: CH is then the channels of ONE lane and a frame has CH * SPLIT channels (8-channel
//           frames as two lanes of 4: every store instruction of a wave is one contiguous 1 KiB)
// PH        1 = the frame has CH * SPLIT - 1 channels (odd totals above 8, SPLIT == 2): the second lane's last channel is a
//           PHANTOM - it multiplies whatever follows the frame in the window and its result is never stored
template <int CH, int TT, int MODE, int NORM, int NTHREADS, int NV, int ASM, int U, int SWZ, int ABL = 0, int OUT16 = 0, int NT = 0, int SPLIT = 1, int PH = 0>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_num_sgpr(CRHIP_SGPR_BUDGET))) void k_poly(const crhip_poly_launch a)
{
	constexpr unsigned CHT = CH * SPLIT - PH;             // channels of a frame
	constexpr unsigned FB = CHT * 2;                      // bytes per input frame (all channels)
	constexpr unsigned FBL = CH * 2;                      // bytes of one lane's share of a frame
	constexpr unsigned TILE_BYTES = NV * 16u * NTHREADS;

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
	volatile unsigned *mailbox = reinterpret_cast<volatile unsigned *>(smem + rows_bytes + 2u * TILE_BYTES);

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
			if (!(ABL == 2 || ABL == 3) || a.n_out == 1)
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
	constexpr int STORES_PER_GROUP = PH ? U * (OUT16 ? CH : stores_of_ints_dword_aligned(CH - 1) + 1)
	                                    : U * (OUT16 ? (CH % 2 == 0 ? stores_of_ints(CH / 2) : CH) : stores_of_ints(CH));

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
	auto draw = [&]() -> unsigned {
		const uint64_t k = (uint64_t)lane_groups + __hip_atomic_fetch_add(lane_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		return k < lane_tiles ? (unsigned)(lane_id + LANES * k) : 0xFFFFFFFFu;
	};
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
	if (dynamic && tid == 0)
		mailbox[0] = draw();
	// stage the polyphase rows once per workgroup (L2-resident after the first workgroups) - AFTER the first tile's DMA and
	// the first ticket are on their way, so that the three round trips of a workgroup's start overlap
	{
		const unsigned nvec = rows_bytes / 16u;
		const u32x4 *src = reinterpret_cast<const u32x4 *>(a.d_rows);
		u32x4 *dst = reinterpret_cast<u32x4 *>(smem);
		for (unsigned i = tid; i < nvec; i += NTHREADS)
			dst[i] = src[i];
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
		const unsigned char *tile = tiles + (it & 1u) * TILE_BYTES;
		const bool more = next_index < n_tiles;
		const uint64_t jn = next_index * NT64;
		unsigned n_next = 0, shift_next = 0;
		unsigned ticket = 0;

		if (more)
		{
			// the other buffer was last read in the previous iteration, which every wave has left (barrier below)
			n_next = (unsigned)((a.n_out - jn < NT64) ? (a.n_out - jn) : NT64);
			shift_next = fetch(jn, n_next, tiles + ((it + 1u) & 1u) * TILE_BYTES);
			if (dynamic && tid == 0)
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
				store_ints_dword_aligned<CH - 1, NT>(out_tile + at, v);
				if (!(L & 1u))
				{
					if constexpr (NT)
						__builtin_nontemporal_store(v[CH - 1], out_tile + at + CH - 1);
					else
						out_tile[at + CH - 1] = v[CH - 1];
				}
			}
		};
		const unsigned char *base = tile + shift + (tid % SPLIT) * FBL;

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
					one_frame<CH, TT, MODE, NORM, ASM, SWZ, SPLIT, PH>(a, rows, base, lane_rel + (first / SPLIT) * a.increment, outv + u * CH);
			}
			if constexpr (ABL == 1 || ABL == 3)
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

		// Full tiles of 4, 2 or 1 groups as straight-line code (see the vmcnt note above).  Specialised instances
		// (TT > 0) run the frames of a tile as a software pipeline: the LDS reads of frame i+1 are issued before the
		// arithmetic of frame i.
		auto run_groups = [&](auto groups_tag) {
			constexpr int G = decltype(groups_tag)::value;
			constexpr int N = G * U;   // frames per lane in this tile

			if constexpr (TT > 0 && (ABL == 0 || ABL == 6))
			{
				FrameData<CH, TT> d[2];
				fetch_frame<CH, TT, MODE, SWZ, SPLIT>(a, rows, base, lane_rel, d[0]);
#pragma unroll
				for (int i = 0; i < N; ++i)
				{
					const unsigned first = (unsigned)(i / U) * GROUP + (unsigned)(i % U) * NTHREADS;   // uniform
					int outv[CH];

					if (i + 1 < N)
					{
						const unsigned next_first = (unsigned)((i + 1) / U) * GROUP + (unsigned)((i + 1) % U) * NTHREADS;
						fetch_frame<CH, TT, MODE, SWZ, SPLIT>(a, rows, base, lane_rel + (next_first / SPLIT) * a.increment, d[(i + 1) & 1]);
					}
					__builtin_amdgcn_sched_barrier(0);   // keep the reads above the arithmetic below
					compute_frame<CH, TT, NORM, ASM>(d[i & 1], outv);

					if constexpr (OUT16)
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
				one_frame<CH, TT, MODE, NORM, ASM, SWZ, SPLIT, PH>(a, rows, base, __umul24(jl / SPLIT, a.increment) + frac0, outv);
				if constexpr (PH)
					store_phantom(jl, outv);
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
	constexpr int STORES_PER_FRAME = OUT16 ? (CH % 2 == 0 ? stores_of_ints(CH / 2) : CH) : stores_of_ints(CH);

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
	const uint64_t global_wave = (uint64_t)blockIdx.x * WAVES + wave;
	const uint64_t global_waves = (uint64_t)gridDim.x * WAVES;

	// tickets: as in k_poly, per wave, 32 counter lanes
	const unsigned LANES = global_waves < 32u ? (unsigned)global_waves : 32u;
	const unsigned lane_id = (unsigned)(global_wave % LANES);
	const uint64_t lane_chunks = n_chunks > lane_id ? (n_chunks - lane_id + LANES - 1u) / LANES : 0;
	const unsigned lane_waves = (unsigned)((global_waves - lane_id + LANES - 1u) / LANES);
	unsigned *lane_counter = a.d_tickets + lane_id * 32u;
	// the draw is split: the atomic is issued at the start of a chunk, its result is first looked at when the last
	// wave-tile of the chunk needs it - by then the per-wave-tile vmcnt waits have long covered it
	auto draw_issue = [&]() -> unsigned {
		unsigned got = 0;
		if (lane == 0)
			got = __hip_atomic_fetch_add(lane_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		return got;
	};
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

// k_generic - reference arithmetic, 64-bit, one lane per output frame (clownresampler.h:986-1035)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_generic(const crhip_generic_launch a)
{
	const uint64_t j = (uint64_t)blockIdx.x * 256u + threadIdx.x;
	if (j >= a.n_out)
		return;

	const uint64_t fr = a.pos_frac + j * a.increment;           // clownresampler.h:1076-1078, j times
	const uint64_t pos_int = a.pos_int + (fr >> 16);
	const uint64_t pos_frac = fr & 0xFFFFu;

	const uint64_t first_rel = (pos_frac + a.delta + 65535u) >> 16;       // :993
	const uint64_t last_rel = (pos_frac + a.skr) >> 16;                   // :994
	const uint64_t first_frame = pos_int + first_rel;                     // :995
	const uint64_t end_frame = pos_int + a.radius_frames + last_rel;      // :996
	uint64_t table_at = (a.step * ((first_rel << 16) - pos_frac)) >> 16;  // :1001

	const short *in = reinterpret_cast<const short *>(a.d_in);
	const unsigned ch = a.channels;

	long long acc[CRHIP_MAX_CHANNELS];
#pragma unroll
	for (int c = 0; c < CRHIP_MAX_CHANNELS; ++c)
		acc[c] = (a.d_acc_in != nullptr && c < (int)ch) ? a.d_acc_in[c] : 0;

	long long weight_sum = 0;

	for (uint64_t f = first_frame; f < end_frame; ++f, table_at += a.step)
	{
		const long long weight = table_at < a.table_len ? a.d_table[table_at] : 0;   // :1012 asserts the index in range
		const short *src = in + f * ch;
		weight_sum += weight;                                                          // :1016
#pragma unroll
		for (int c = 0; c < CRHIP_MAX_CHANNELS; ++c)
			if (c < (int)ch)
				acc[c] += (long long)src[c] * weight / 65536;                          // :1020
	}

	// :1025 - the host refuses configurations whose weight sum can be 0 (the reference traps there)
	const long long reciprocal = weight_sum != 0 ? 2147483648ll / weight_sum : 0;

#pragma unroll
	for (int c = 0; c < CRHIP_MAX_CHANNELS; ++c)
	{
		if (c < (int)ch)
		{
			const long long v = acc[c] * reciprocal / 32768;                           // :1033
			if (a.out64 == 2)
				reinterpret_cast<short *>(a.d_out)[j * ch + c] = (short)(v > 0x7FFF ? 0x7FFF : (v < -0x7FFF ? -0x7FFF : v));
			else if (a.out64)
				reinterpret_cast<long long *>(a.d_out)[j * ch + c] = v;
			else
				reinterpret_cast<int *>(a.d_out)[j * ch + c] = (int)v;
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// Instance table of k_poly
// ---------------------------------------------------------------------------------------------------------
typedef void (*poly_fn)(const crhip_poly_launch);

// Tuning variants of the specialised instances: geometry x frames in flight x non-temporal stores.
// (A swizzled LDS row image, SWZ = 1, measured no better than the plain one and is not instantiated.)
//   variant = geo + 5 * ui + 10 * nt     geo: 0 (256 thr, 2 vec) 1 (512,1) 2 (512,2) 3 (1024,1) 4 (1024,2); ui: 0/1 -> U = 1/2
// All k_poly instances use the SDWA arithmetic (ASM = 1); the plain-C form (ASM = 0) is kept in the source as its
// readable definition, and k_generic is the independent 64-bit implementation the tests compare against the oracle too.
struct geometry
{
	int threads, vecs;
};
constexpr geometry GEOMETRY[5] = {{256, 2}, {512, 1}, {512, 2}, {1024, 1}, {1024, 2}};
constexpr int VARIANTS = 20;

template <int CH, int TT, int MODE, int NORM, int GEO, int ASM, int UI, int NT, int OUT16 = 0>
constexpr poly_fn instance()
{
	return (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[GEO].threads, GEOMETRY[GEO].vecs, ASM, (1 << UI), 0, 0, OUT16, NT>;
}

template <int CH, int TT, int MODE, int NORM, int V>
struct variant_table
{
	static void fill(poly_fn *t)
	{
		t[V] = instance<CH, TT, MODE, NORM, V % 5, 1, (V / 5) % 2, (V / 10) % 2>();
		variant_table<CH, TT, MODE, NORM, V + 1>::fill(t);
	}
};
template <int CH, int TT, int MODE, int NORM>
struct variant_table<CH, TT, MODE, NORM, VARIANTS>
{
	static void fill(poly_fn *) {}
};

// specialised (channels, slots, mode, norm) instances; the BASELINE.json configurations
struct special
{
	uint32_t channels, slots, mode, norm;
	uint32_t default_variant;   // from tools/sweep_variants.py on MI355X (profiles/)
	poly_fn fn[VARIANTS];
	poly_fn fn16;               // int16-output form, default variant only
	poly_fn wave[2];            // k_wave (variants WAVE_VARIANT + {0: non-temporal stores, 1: plain}); nullptr if none
	poly_fn wave16;             // k_wave, int16 output
	bool dynamic_tiles;         // k_poly: draw tiles as tickets (measured per instance; see crhip_poly_launch.dynamic_tiles)
	poly_fn split[4];           // k_poly with two lanes per frame (variants SPLIT_VARIANT + i: geometry {4, 2} x nt {1, 0}); nullptr if none
	poly_fn up[2];              // k_up (variants UP_VARIANT + {0: 24-bit multiply-add + SDWA add per tap, 1: 64-bit multiply-add chain}); nullptr if none
	poly_fn up16;               // k_up, int16 output
	bool lite;                  // one k_poly instance only (the default variant, int32 and int16 forms): every variant id resolves to it
	uint32_t lite_lanes;        // lite instances: lanes per frame (2: each lane takes half of the channels, as the run-time instances above 8 channels do)
	uint32_t up_negmask;        // k_up / mad: bit s set = the weights of slot s are <= 0 in every row, clear = >= 0 (checked by the host per plan)
	poly_fn mad[2];             // the 64-bit multiply-add chain (compute_frame, ASM mode 2): variant MAD_VARIANT = k_poly geometry 3 with non-temporal stores,
	                            // MAD_VARIANT + 1 = k_wave where the instance has one, else k_poly geometry 3 with plain stores
};

constexpr uint32_t MAD_VARIANT = 28;    // variant ids 28, 29

constexpr uint32_t UP_VARIANT = 26;     // variant ids 26, 27 select k_up where the instance has one and the plan qualifies
constexpr int UP_WAVES = 12;
constexpr uint32_t UP_MAX_WAVE_TILE = 1024;   // output frames per wave-tile (LDS staging)

constexpr uint32_t SPLIT_VARIANT = 22;   // variant ids 22..25
constexpr int SPLIT_GEO[4] = {4, 2, 4, 2};
constexpr int SPLIT_NT[4] = {1, 1, 0, 0};

constexpr uint32_t WAVE_VARIANT = 20;   // variant ids 20, 21 select k_wave where the instance has one
constexpr int WAVE_WAVES = 16, WAVE_NVW = 1, WAVE_ITER = 4;

template <int CH, int TT, int MODE, int NORM, int DV, bool WAVE = false, bool DYNAMIC = false, unsigned UPMASK = 0>
special make_special()
{
	special s = {CH, TT, MODE, NORM, DV, {}, nullptr, {nullptr, nullptr}, nullptr, DYNAMIC, {nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr}, nullptr, false, 1u, UPMASK, {nullptr, nullptr}};
	if constexpr (UPMASK != 0 && CH % 2 == 0)
	{
		s.mad[0] = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), 1, 0, 0, 0, 1>;
		s.mad[1] = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), 1, 0, 0, 0, 0>;
		if constexpr (WAVE)
			s.mad[1] = (poly_fn)k_wave<CH, TT, MODE, NORM, WAVE_WAVES, WAVE_NVW, WAVE_ITER, 0, 1, 0, (int)(2u | (UPMASK << 8))>;   // where the instance has a k_wave form, 29 is k_wave with the chain
	}
	if constexpr (UPMASK != 0)
	{
		static_assert(MODE == CRHIP_ROWMODE_UPSAMPLE, "k_up is for pure upsampling");
		s.up[0] = (poly_fn)k_up<CH, TT, NORM, UPMASK, UP_WAVES, 0, 1>;
		s.up[1] = (poly_fn)k_up<CH, TT, NORM, UPMASK, UP_WAVES, 0, 1, 0, 1>;   // the 64-bit chain form
		s.up16 = (poly_fn)k_up<CH, TT, NORM, UPMASK, UP_WAVES, 1, 1, 0, 1>;
	}
	variant_table<CH, TT, MODE, NORM, 0>::fill(s.fn);
	constexpr int KV = DV < 20 ? DV : 13;   // the k_poly variant behind a k_wave default (its fallback and int16 geometry)
	s.fn16 = instance<CH, TT, MODE, NORM, KV % 5, 1, (KV / 5) % 2, (KV / 10) % 2, 1>();
	if constexpr (CH % 2 == 0 && CH >= 8)
	{
		// two lanes per frame, each with CH / 2 channels
		s.split[0] = (poly_fn)k_poly<CH / 2, TT, MODE, NORM, GEOMETRY[4].threads, GEOMETRY[4].vecs, 1, 1, 0, 0, 0, 1, 2>;
		s.split[1] = (poly_fn)k_poly<CH / 2, TT, MODE, NORM, GEOMETRY[2].threads, GEOMETRY[2].vecs, 1, 1, 0, 0, 0, 1, 2>;
		s.split[2] = (poly_fn)k_poly<CH / 2, TT, MODE, NORM, GEOMETRY[4].threads, GEOMETRY[4].vecs, 1, 1, 0, 0, 0, 0, 2>;
		s.split[3] = (poly_fn)k_poly<CH / 2, TT, MODE, NORM, GEOMETRY[2].threads, GEOMETRY[2].vecs, 1, 1, 0, 0, 0, 0, 2>;
	}
	if constexpr (WAVE)
	{
		s.wave[0] = (poly_fn)k_wave<CH, TT, MODE, NORM, WAVE_WAVES, WAVE_NVW, WAVE_ITER, 0, 1>;
		s.wave[1] = (poly_fn)k_wave<CH, TT, MODE, NORM, WAVE_WAVES, WAVE_NVW, WAVE_ITER, 0, 0>;
		s.wave16 = (poly_fn)k_wave<CH, TT, MODE, NORM, WAVE_WAVES, WAVE_NVW, WAVE_ITER, 1, 1>;
	}
	return s;
}

// A specialised instance WITHOUT the tuning variants: one k_poly (compile-time slot count, pipelined LDS reads) at the
// geometry the run-time-slot instance of that channel count uses, non-temporal stores, int32 and int16 forms.  For the common
// surround layouts, where the run-time-slot loop leaves 10-20 % behind (profiles/r01_channel_table.log).
template <int CH, int TT, int MODE, int NORM, int DV = (CH <= 4 ? 13 : 14)>   // default: (1024 threads, 1 or 2 vectors per thread), one frame in flight, non-temporal stores
special make_special_lite()
{
	special s = {CH, TT, MODE, NORM, DV, {}, nullptr, {nullptr, nullptr}, nullptr, false, {nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr}, nullptr, true, 1u, 0u, {nullptr, nullptr}};
	const poly_fn fn = instance<CH, TT, MODE, NORM, DV % 5, 1, (DV / 5) % 2, (DV / 10) % 2>();
	for (int v = 0; v < VARIANTS; ++v)
		s.fn[v] = fn;
	s.fn16 = instance<CH, TT, MODE, NORM, DV % 5, 1, (DV / 5) % 2, (DV / 10) % 2, 1>();
	return s;
}

// ... and with two lanes per frame (CHT channels in all, CHT / 2 per lane), at the geometry of the run-time instances above 8 channels
template <int CHT, int TT, int MODE, int NORM>
special make_special_lite_split()
{
	constexpr int DV = 14;   // (1024 threads, 2 vectors per thread), one frame in flight, non-temporal stores
	constexpr int T = GEOMETRY[DV % 5].threads, V = GEOMETRY[DV % 5].vecs;
	special s = {CHT, TT, MODE, NORM, DV, {}, nullptr, {nullptr, nullptr}, nullptr, false, {nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr}, nullptr, true, 2u, 0u, {nullptr, nullptr}};
	const poly_fn fn = (poly_fn)k_poly<CHT / 2, TT, MODE, NORM, T, V, 1, 1, 0, 0, 0, 1, 2>;
	for (int v = 0; v < VARIANTS; ++v)
		s.fn[v] = fn;
	s.fn16 = (poly_fn)k_poly<CHT / 2, TT, MODE, NORM, T, V, 1, 1, 0, 0, 1, 1, 2>;
	return s;
}

const special *specials(int *count)
{
	static const special table[] = {
	    // the k_up sign masks are those of a Lanczos window whose lobes are one input frame wide (slot 0 = first_slot)
	    make_special<2, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 13, true, true, 0x12u>(),   // cfg 2 / cfg 5: stereo 44.1 -> 48 kHz, 3 lobes
	    make_special<2, 15, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32, 27, true, false, 0x2A55u>(),  // cfg 3: stereo 8 -> 96 kHz, 8 lobes (k_up, chain form, from 2x upsampling on; k_wave below)
	    make_special<8, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 2>(),     // cfg 4: 8 channels 48 -> 44.1 kHz (5-6 taps; 6 slots on shifted windows)
	    make_special<1, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 13, true, true>(),   // mono upsampling, 3 lobes
	    make_special<2, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 13, false, true>(),     // stereo mild downsampling, 3 lobes
	    make_special<1, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 13, false, true>(),     // mono mild downsampling, 3 lobes
	    make_special_lite<4, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31>(),               // quad, 5.1 and 7.1 at 44.1 <-> 48 kHz
	    make_special_lite<4, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite<6, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31>(),
	    make_special_lite<6, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite<8, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 2>(),            // (the geometry cfg 4's instance measured best with: 512 threads, plain stores)
	    make_special_lite<3, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31>(),               // and the odd layouts in between (2.1, 5.0, 6.1)
	    make_special_lite<3, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite<5, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31>(),
	    make_special_lite<7, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31>(),
	    make_special_lite<7, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite_split<12, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31>(),        // 7.1.4 and 16 channels (the reference's maximum) at 44.1 <-> 48 kHz; (16,5) measured SLOWER than the run-time instance
	    make_special_lite_split<12, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite_split<16, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    // mono and stereo at the usual downsampling ratios: 2:1 (12 slots), 96 -> 44.1 (13), 3:2 (9), 44.1 -> 32 (8), 3:1 (18)
	    make_special_lite<1, 12, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite<2, 12, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite<1, 13, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite<2, 13, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite<1, 9, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite<2, 9, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite<1, 8, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite<2, 8, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite<1, 18, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite<2, 18, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    // the 8-lobe build: mono upsampling, mono / stereo 48 -> 44.1 kHz (17 slots)
	    make_special_lite<1, 15, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32>(),
	    make_special_lite<1, 17, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite<2, 17, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	};
	*count = (int)(sizeof(table) / sizeof(table[0]));
	return table;
}

// timing-only ablations of the headline instance at the default geometry (see ABL above)
poly_fn ablation_instance(int abl)
{
	switch (abl)
	{
		case 1: return (poly_fn)k_poly<2, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 1024, 1, 1, 1, 0, 1>;
		case 2: return (poly_fn)k_poly<2, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 1024, 1, 1, 1, 0, 2>;
		case 3: return (poly_fn)k_poly<2, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 1024, 1, 1, 1, 0, 3>;
		case 4: return (poly_fn)k_poly<2, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 1024, 1, 1, 1, 0, 4>;
		case 5: return (poly_fn)k_poly<2, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 1024, 1, 1, 1, 0, 4, 0, 1>;   // as 4, non-temporal stores
		case 6: return (poly_fn)k_poly<2, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 1024, 1, 1, 1, 0, 6, 0, 1>;   // the real kernel (variant 13) + clock stamps
		case 7: return (poly_fn)k_wave<2, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, WAVE_WAVES, WAVE_NVW, WAVE_ITER, 0, 1, 6>;   // k_wave + stamps
		case 8: return (poly_fn)k_up<2, 15, CRHIP_NORM_U32, 0x2A55u, UP_WAVES, 0, 1, 6, 1>;   // k_up of the 8-lobe stereo instance (chain form, the default) + stamps
		default: return nullptr;
	}
}

const special *find_special(uint32_t channels, uint32_t slots, uint32_t mode, uint32_t norm)
{
	int n;
	const special *t = specials(&n);
	for (int i = 0; i < n; ++i)
		if (t[i].channels == channels && t[i].slots == slots && t[i].mode == mode && t[i].norm == norm)
			return &t[i];
	return nullptr;
}

// run-time slot count: every channel count 1..8 and the even counts 10..16 (the reference's maximum,
// CLOWNRESAMPLER_MAXIMUM_CHANNELS, clownresampler.h:462), both row modes, both normalisations, both output forms.  One
// geometry per channel count - 1024 threads; 16 KiB tiles for up to 4 channels, 32 KiB above (an 8-channel frame is 16
// bytes) - SDWA arithmetic, one frame in flight, non-temporal stores.  Above 8 channels a frame is shared by TWO neighbouring
// lanes (k_poly's SPLIT), each taking half of its channels: the per-lane code is that of 5..8 channels.
constexpr int runtime_geo(int channels)
{
	return channels <= 4 ? 3 : 4;   // (512 threads x 2 vectors, which the specialised 8-channel instances prefer, measured 10-20 % slower here)
}
constexpr int runtime_split(int channels)
{
	return channels > 8 ? 2 : 1;
}

template <int CH, int OUT16>
poly_fn pick_runtime(uint32_t mode, uint32_t norm)
{
	constexpr int GEO = runtime_geo(CH);
	if (norm == CRHIP_NORM_S31)
		return mode == CRHIP_ROWMODE_UPSAMPLE ? instance<CH, 0, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, GEO, 1, 0, 1, OUT16>()
		                                      : instance<CH, 0, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, GEO, 1, 0, 1, OUT16>();
	return mode == CRHIP_ROWMODE_UPSAMPLE ? instance<CH, 0, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32, GEO, 1, 0, 1, OUT16>()
	                                      : instance<CH, 0, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_U32, GEO, 1, 0, 1, OUT16>();
}

// two lanes per frame, HALF channels each (run-time slot count, geometry 4); PH = 1: 2 * HALF - 1 channels (see k_poly)
template <int HALF, int OUT16, int PH = 0>
poly_fn pick_runtime_split(uint32_t mode, uint32_t norm)
{
	constexpr int T = GEOMETRY[runtime_geo(16)].threads, V = GEOMETRY[runtime_geo(16)].vecs;
	if (norm == CRHIP_NORM_S31)
		return mode == CRHIP_ROWMODE_UPSAMPLE ? (poly_fn)k_poly<HALF, 0, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, T, V, 1, 1, 0, 0, OUT16, 1, 2, PH>
		                                      : (poly_fn)k_poly<HALF, 0, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, T, V, 1, 1, 0, 0, OUT16, 1, 2, PH>;
	return mode == CRHIP_ROWMODE_UPSAMPLE ? (poly_fn)k_poly<HALF, 0, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32, T, V, 1, 1, 0, 0, OUT16, 1, 2, PH>
	                                      : (poly_fn)k_poly<HALF, 0, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_U32, T, V, 1, 1, 0, 0, OUT16, 1, 2, PH>;
}

template <int OUT16>
poly_fn pick_runtime_channels(uint32_t channels, uint32_t mode, uint32_t norm)
{
	switch (channels)
	{
		case 1: return pick_runtime<1, OUT16>(mode, norm);
		case 2: return pick_runtime<2, OUT16>(mode, norm);
		case 3: return pick_runtime<3, OUT16>(mode, norm);
		case 4: return pick_runtime<4, OUT16>(mode, norm);
		case 5: return pick_runtime<5, OUT16>(mode, norm);
		case 6: return pick_runtime<6, OUT16>(mode, norm);
		case 7: return pick_runtime<7, OUT16>(mode, norm);
		case 8: return pick_runtime<8, OUT16>(mode, norm);
		case 9: return pick_runtime_split<5, OUT16, 1>(mode, norm);
		case 10: return pick_runtime_split<5, OUT16>(mode, norm);
		case 11: return pick_runtime_split<6, OUT16, 1>(mode, norm);
		case 13: return pick_runtime_split<7, OUT16, 1>(mode, norm);
		case 15: return pick_runtime_split<8, OUT16, 1>(mode, norm);
		case 12: return pick_runtime_split<6, OUT16>(mode, norm);
		case 14: return pick_runtime_split<7, OUT16>(mode, norm);
		case 16: return pick_runtime_split<8, OUT16>(mode, norm);
		default: return nullptr;
	}
}

} // namespace

// -------------------------------------------------------------------------------------------------------------
// C-ABI shim
// -------------------------------------------------------------------------------------------------------------
extern "C"
{

const char *crhip_error_string(int code)
{
	return hipGetErrorString((hipError_t)code);
}

int crhip_device_count(int *count)
{
	*count = 0;
	return (int)hipGetDeviceCount(count);
}

int crhip_set_device(int ordinal)
{
	return (int)hipSetDevice(ordinal);
}

int crhip_get_device_info(int ordinal, crhip_device_info *info)
{
	hipDeviceProp_t prop;
	const hipError_t e = hipGetDeviceProperties(&prop, ordinal);
	if (e != hipSuccess)
		return (int)e;
	memset(info, 0, sizeof(*info));
	info->compute_units = prop.multiProcessorCount;
	info->max_lds_per_block = (int)prop.sharedMemPerBlock;
	info->wavefront = prop.warpSize;
	info->clock_khz = prop.clockRate;
	info->total_memory = prop.totalGlobalMem;
	strncpy(info->name, prop.name, sizeof(info->name) - 1);
	strncpy(info->arch, prop.gcnArchName, sizeof(info->arch) - 1);
	return 0;
}

int crhip_malloc(void **device_pointer, size_t bytes)
{
	return (int)hipMalloc(device_pointer, bytes);
}

int crhip_free(void *device_pointer)
{
	return (int)hipFree(device_pointer);
}

int crhip_host_alloc(void **host_pointer, size_t bytes)
{
	return (int)hipHostMalloc(host_pointer, bytes, hipHostMallocDefault);
}

int crhip_host_free(void *host_pointer)
{
	return (int)hipHostFree(host_pointer);
}

int crhip_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
	return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
}

int crhip_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
	return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream);
}

int crhip_memset(void *dst, int value, size_t bytes, void *stream)
{
	return (int)hipMemsetAsync(dst, value, bytes, (hipStream_t)stream);
}

int crhip_stream_create(void **stream)
{
	hipStream_t s = nullptr;
	const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	*stream = (void *)s;
	return (int)e;
}

int crhip_stream_destroy(void *stream)
{
	return (int)hipStreamDestroy((hipStream_t)stream);
}

int crhip_stream_sync(void *stream)
{
	return (int)hipStreamSynchronize((hipStream_t)stream);
}

int crhip_device_sync(void)
{
	return (int)hipDeviceSynchronize();
}

int crhip_event_create(void **event)
{
	hipEvent_t e = nullptr;
	const hipError_t r = hipEventCreateWithFlags(&e, hipEventDisableTiming);
	*event = (void *)e;
	return (int)r;
}

int crhip_event_destroy(void *event)
{
	return (int)hipEventDestroy((hipEvent_t)event);
}

int crhip_event_record(void *event, void *stream)
{
	return (int)hipEventRecord((hipEvent_t)event, (hipStream_t)stream);
}

int crhip_stream_wait_event(void *stream, void *event)
{
	return (int)hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0);
}

int crhip_stream_is_capturing(void *stream, int *capturing)
{
	hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
	const hipError_t e = hipStreamIsCapturing((hipStream_t)stream, &status);
	*capturing = (e == hipSuccess && status == hipStreamCaptureStatusActive) ? 1 : 0;
	return (int)e;
}

int crhip_stream_busy(void *stream)
{
	const hipError_t e = hipStreamQuery((hipStream_t)stream);
	if (e == hipErrorNotReady)
		return 1;
	if (e != hipSuccess)
		(void)hipGetLastError();   // a handle the caller has destroyed: not an error of ours, and certainly not busy
	return 0;
}

int crhip_enable_peer_access(int device, int peer)
{
	int can = 0, current = 0;
	hipError_t e = hipGetDevice(&current);
	if (e != hipSuccess)
		return (int)e;
	if (device == peer || hipDeviceCanAccessPeer(&can, device, peer) != hipSuccess || !can)
		return 0;
	e = hipSetDevice(device);
	if (e == hipSuccess)
	{
		e = hipDeviceEnablePeerAccess(peer, 0);
		if (e == hipErrorPeerAccessAlreadyEnabled)
		{
			(void)hipGetLastError();
			e = hipSuccess;
		}
	}
	(void)hipSetDevice(current);
	return (int)e;
}

int crhip_memcpy_peer(void *dst, int dst_device, const void *src, int src_device, size_t bytes, void *stream)
{
	if (dst_device == src_device)
		return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
	return (int)hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, (hipStream_t)stream);
}

int crhip_get_device(int *ordinal)
{
	return (int)hipGetDevice(ordinal);
}

int crhip_poly_has_instance(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode)
{
	return find_special(channels, slots, row_mode, norm_mode) != nullptr ? 1 : 0;
}

// variant actually used by a launch: an explicit one (< VARIANTS), an ablation (>= 1000: headline geometry), or the
// instance's measured default (CRHIP_VARIANT_DEFAULT)
static uint32_t resolve_variant(const special *sp, uint32_t variant, uint32_t out_s16 = 0)
{
	if (sp == nullptr)
		return 0u;
	if (sp->lite)
		return sp->default_variant;
	if (variant == 1008u)
		return sp->up[0] != nullptr ? UP_VARIANT : (sp->wave[0] != nullptr ? WAVE_VARIANT : 13u);   // diagnostic k_up instance
	if (variant == 1007u && sp->wave[0] != nullptr)
		return WAVE_VARIANT;                                  // diagnostic k_wave instance: k_wave geometry
	if (variant >= 1000u && variant < 1010u)
		return 3u;                                            // diagnostic k_poly instances: headline geometry
	if (variant >= MAD_VARIANT + 2u)
		variant = sp->default_variant;
	if (variant >= MAD_VARIANT)
	{
		if (sp->mad[0] != nullptr && !out_s16)
			return variant;
		variant = 13u;
	}
	if (variant >= UP_VARIANT)
	{
		if (sp->up[0] != nullptr)
			return variant;
		variant = sp->wave[0] != nullptr ? WAVE_VARIANT : 13u;
	}
	if (variant >= SPLIT_VARIANT && sp->split[0] == nullptr)
		variant = 13u;
	if (variant >= WAVE_VARIANT && variant < SPLIT_VARIANT && sp->wave[0] == nullptr)
		variant = 13u;
	if (out_s16 && variant >= SPLIT_VARIANT)
		variant = sp->default_variant < WAVE_VARIANT ? sp->default_variant : 13u;
	if (out_s16 && variant < WAVE_VARIANT)
		return sp->default_variant < WAVE_VARIANT ? sp->default_variant : 13u;   // the k_poly int16 form exists for one variant
	return variant;
}

int crhip_poly_swizzled(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode, uint32_t variant)
{
	const special *sp = find_special(channels, slots, row_mode, norm_mode);
	(void)sp;
	(void)variant;
	return 0;   // no swizzled instance is built
}

int crhip_poly_variants(void)
{
	return VARIANTS + 10;   // + the two k_wave variants, the four two-lanes-per-frame variants, the two k_up variants and the two 64-bit-chain variants
}

int crhip_poly_up_negmask(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode, uint32_t *negmask)
{
	const special *sp = find_special(channels, slots, row_mode, norm_mode);
	if (sp == nullptr || (sp->up[0] == nullptr && sp->mad[0] == nullptr))
		return 0;
	*negmask = sp->up_negmask;
	return 1;
}

int crhip_poly_default_is_mad(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode)
{
	const special *sp = find_special(channels, slots, row_mode, norm_mode);
	return sp != nullptr && sp->default_variant >= MAD_VARIANT && sp->default_variant < MAD_VARIANT + 2u && sp->mad[0] != nullptr ? 1 : 0;
}

uint32_t crhip_poly_up_fallback_variant(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode)
{
	const special *sp = find_special(channels, slots, row_mode, norm_mode);
	return sp != nullptr && sp->wave[0] != nullptr ? WAVE_VARIANT : 13u;
}

int crhip_poly_dynamic_default(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode)
{
	const special *sp = find_special(channels, slots, row_mode, norm_mode);
	return sp != nullptr && sp->dynamic_tiles ? 1 : 0;
}

uint32_t crhip_poly_fallback_variant(void)
{
	return 13u;   // a k_poly variant every specialised instance has
}

void crhip_poly_geometry(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode, uint32_t variant,
                         uint32_t *threads, uint32_t *vecs, uint32_t *frames_multiple)
{
	const special *sp = find_special(channels, slots, row_mode, norm_mode);
	const uint32_t v = sp != nullptr ? resolve_variant(sp, variant) : (uint32_t)runtime_geo((int)channels);

	if (sp != nullptr && v == MAD_VARIANT + 1u && sp->wave[0] != nullptr)
	{
		*threads = WAVE_WAVES * 64u;
		*vecs = 100u + WAVE_NVW;
		*frames_multiple = 64u * WAVE_ITER * 4u;
		return;
	}

	if (sp != nullptr && v >= MAD_VARIANT)
	{
		*threads = (uint32_t)GEOMETRY[3].threads;
		*vecs = (uint32_t)GEOMETRY[3].vecs;
		*frames_multiple = *threads;
		return;
	}

	if (sp != nullptr && v >= UP_VARIANT)
	{
		// k_up: vecs = 200; frames_multiple = the cap on output frames per wave-tile (the host sizes the wave-tile so that
		// it spans at most 64 input positions)
		*threads = UP_WAVES * 64u;
		*vecs = 200u;
		*frames_multiple = UP_MAX_WAVE_TILE;
		return;
	}

	if (sp != nullptr && v >= SPLIT_VARIANT)
	{
		// two lanes per frame: a group of threads covers half as many frames
		*threads = (uint32_t)GEOMETRY[SPLIT_GEO[v - SPLIT_VARIANT]].threads;
		*vecs = (uint32_t)GEOMETRY[SPLIT_GEO[v - SPLIT_VARIANT]].vecs;
		*frames_multiple = *threads / 2u;
		return;
	}

	if (sp != nullptr && v >= WAVE_VARIANT)
	{
		// k_wave: vecs = 100 + (1 KiB DMA pieces per wave-tile); frames_multiple = frames per ticket
		*threads = WAVE_WAVES * 64u;
		*vecs = 100u + WAVE_NVW;
		*frames_multiple = 64u * WAVE_ITER * 4u;
		return;
	}

	*threads = (uint32_t)GEOMETRY[v % 5].threads;
	*vecs = (uint32_t)GEOMETRY[v % 5].vecs;
	*frames_multiple = *threads * (1u << ((v / 5) % 2));
	if (sp == nullptr)
		*frames_multiple /= (uint32_t)runtime_split((int)channels);   // a group of threads covers half as many frames
	else if (sp->lite)
		*frames_multiple /= sp->lite_lanes;
}

// geo: index into GEOMETRY, 100 for k_wave, 200 for k_up
static poly_fn select_poly(const crhip_poly_launch *launch, uint32_t *geo)
{
	const special *sp = launch->specialised ? find_special(launch->channels, launch->slots, launch->row_mode, launch->norm_mode) : nullptr;
	const uint32_t v = resolve_variant(sp, launch->variant, launch->out_s16);
	poly_fn fn;

	if (sp != nullptr && v >= MAD_VARIANT)
	{
		*geo = (v == MAD_VARIANT + 1u && sp->wave[0] != nullptr) ? 100u : 3u;
		return sp->mad[v - MAD_VARIANT];
	}

	if (sp != nullptr && v >= UP_VARIANT)
	{
		*geo = 200u;
		if (launch->variant == 1008u && !launch->out_s16 && launch->channels == 2 && launch->slots == 15)
			return ablation_instance(8);
		return launch->out_s16 ? sp->up16 : sp->up[v - UP_VARIANT];
	}

	if (sp != nullptr && v >= SPLIT_VARIANT)
	{
		*geo = (uint32_t)SPLIT_GEO[v - SPLIT_VARIANT];
		return sp->split[v - SPLIT_VARIANT];
	}

	if (sp != nullptr && v >= WAVE_VARIANT)
	{
		fn = launch->out_s16 ? sp->wave16 : sp->wave[v - WAVE_VARIANT];
		if (launch->variant == 1007u)
			fn = ablation_instance(7);
		*geo = 100u;
		return fn;
	}

	if (launch->out_s16)
		fn = sp != nullptr ? sp->fn16 : pick_runtime_channels<1>(launch->channels, launch->row_mode, launch->norm_mode);
	else
		fn = sp != nullptr ? sp->fn[v] : pick_runtime_channels<0>(launch->channels, launch->row_mode, launch->norm_mode);

	// debug: variant 1000 + k selects timing-only ablation k of the headline instance (results are wrong by design)
	if (sp != nullptr && launch->variant >= 1000u && launch->variant < 1010u && launch->channels == 2 && launch->slots == 5)
		fn = ablation_instance((int)(launch->variant - 1000u));

	*geo = sp != nullptr ? v % 5 : (uint32_t)runtime_geo((int)launch->channels);
	return fn;
}

// One-time per-instance setup (not legal inside a stream capture): allow more than 48 KiB of dynamic LDS.
int crhip_poly_prepare(const crhip_poly_launch *launch)
{
	uint32_t geo;
	const poly_fn fn = select_poly(launch, &geo);

	if (fn == nullptr)
		return (int)hipErrorInvalidValue;

	// hipFuncSetAttribute costs ~0.2 ms: remember, per device, the largest size each instance has been opened up to, so that
	// a client walking through many ratios (a new plan each) pays it once per instance
	{
		static std::mutex lock;
		static std::map<std::pair<int, const void *>, uint32_t> opened;
		int device = 0;
		const hipError_t e = hipGetDevice(&device);
		if (e != hipSuccess)
			return (int)e;

		std::lock_guard<std::mutex> guard(lock);
		uint32_t &bytes = opened[std::make_pair(device, (const void *)fn)];
		if (bytes >= launch->lds_bytes)
			return 0;
		const hipError_t r = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)launch->lds_bytes);
		if (r == hipSuccess)
			bytes = launch->lds_bytes;
		return (int)r;
	}
}

// What the runtime says about the instance a launch would use: workgroups of launch->threads threads and launch->lds_bytes of
// dynamic LDS that fit one CU at a time, and the instance's register / static LDS footprint.
int crhip_poly_occupancy(const crhip_poly_launch *launch, int *workgroups_per_cu, int *vgprs, int *static_lds)
{
	uint32_t geo;
	const poly_fn fn = select_poly(launch, &geo);
	hipFuncAttributes attr;
	hipError_t e;

	if (fn == nullptr)
		return (int)hipErrorInvalidValue;
	e = hipFuncGetAttributes(&attr, (const void *)fn);
	if (e != hipSuccess)
		return (int)e;
	*vgprs = attr.numRegs;
	*static_lds = (int)attr.sharedSizeBytes;
	return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(workgroups_per_cu, (const void *)fn, (int)launch->threads, launch->lds_bytes);
}

int crhip_launch_poly(const crhip_poly_launch *launch, void *stream)
{
	uint32_t geo;
	const poly_fn fn = select_poly(launch, &geo);

	if (fn == nullptr)
		return (int)hipErrorInvalidValue;
	if (geo == 200u ? (launch->threads != UP_WAVES * 64u || launch->vecs != 200u || launch->tile_frames % 4u != 0 || launch->tile_frames / 4u > UP_MAX_WAVE_TILE || launch->plane_rows != UP_PLANE_ROWS)
	  : geo == 100u ? (launch->threads != WAVE_WAVES * 64u || launch->vecs != 100u + WAVE_NVW)
	                : (launch->threads != (uint32_t)GEOMETRY[geo].threads || launch->vecs != (uint32_t)GEOMETRY[geo].vecs))
		return (int)hipErrorInvalidValue;
	if (launch->n_out == 0)
		return 0;

	hipLaunchKernelGGL(fn, dim3(launch->blocks), dim3(launch->threads), launch->lds_bytes, (hipStream_t)stream, *launch);
	return (int)hipGetLastError();
}

int crhip_launch_generic(const crhip_generic_launch *launch, void *stream)
{
	if (launch->n_out == 0)
		return 0;
	if (launch->channels == 0 || launch->channels > CRHIP_MAX_CHANNELS)
		return (int)hipErrorInvalidValue;

	const uint64_t blocks = (launch->n_out + 255u) / 256u;
	if (blocks > 0x7FFFFFFFull)
		return (int)hipErrorInvalidValue;

	hipLaunchKernelGGL(k_generic, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *launch);
	return (int)hipGetLastError();
}

} // extern "C"
