// chainbench.hip - what the frame loop of the input-stationary kernel (k_up / k_up2, cr_kup.hpp) costs per wave-frame on
// MI355X, form by form, at the kernel's own occupancy (one 768-thread workgroup per CU = 3 waves per SIMD; 4 for comparison).
// A wave-frame here is: 4 x ds_read_b128 of a weight row (prefetched one frame ahead), 15 taps x 2 channels of the tap
// arithmetic in the selected FORM, the 64-bit normalisation, one ds_write_b64 of the result.  Results are not meaningful
// numbers; only the time is.  Prints cycles per wave-frame per SIMD (from the in-kernel clock) for every form.
//
//   FORM 0  v_ashrrev lo + v_mad_i64_i32, 2 accumulator pairs (one per channel), one asm statement per tap and channel
//   FORM 1  the same with 4 accumulator pairs (even / odd slots per channel)
//   FORM 2  as 0, but the 30 taps of a frame in ONE asm statement (no padding between statements)
//   FORM 3  as 1, one asm statement per frame
//   FORM 4  round 1's form: v_mov lo, bias + v_mad_i64_i32, 4 pairs, one statement per tap
//   FORM 5  v_mad_i32_i24 + v_add_u32_sdwa (the 24-bit form), one statement per slot
//   FORM 6  as 3 without the LDS row reads (weights stay in registers): what the LDS traffic costs
//   FORM 7  as 3 without the staging store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

constexpr int TT = 15;
constexpr unsigned PLANE_ROWS = 1040;

template <int FORM, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(int *out, int frames, unsigned increment, unsigned long long *stamps)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	// rows: 4 planes x PLANE_ROWS x 16 bytes of small positive weights
	for (unsigned i = tid; i < 4u * PLANE_ROWS * 4u; i += WAVES * 64u)
		reinterpret_cast<int *>(smem)[i] = (int)((i * 2654435761u) >> 16) & 0x1FFFF;
	__syncthreads();
	unsigned char *stage = smem + 4u * PLANE_ROWS * 16u + wave * (WAVES > 12 ? 4096u : 6144u);

	int S[TT][2], B[TT][2];
#pragma unroll
	for (int s = 0; s < TT; ++s)
	{
		S[s][0] = (int)((lane * 977u + s * 131u + 7u) << 15) >> 1;
		S[s][1] = -(int)((lane * 613u + s * 257u + 3u) << 14) >> 1;
		B[s][0] = S[s][0] >> 31;
		B[s][1] = S[s][1] >> 31;
		asm volatile("" : "+v"(S[s][0]), "+v"(S[s][1]), "+v"(B[s][0]), "+v"(B[s][1]));
	}

	const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	unsigned g = 65536u - ((lane * 4u) & 0xFFFu);
	unsigned stage_at = lane * 96u;
	int wa[16], wb[16];
	auto read_row = [&](unsigned gg, int (&w)[16]) {
		if constexpr (FORM == 6)
		{
#pragma unroll
			for (int q = 0; q < 16; ++q)
				asm volatile("" : "+v"(w[q]));
			return;
		}
		const unsigned at = (gg >> 2) & 0x7FF0u;
		const i32x4 *plane0 = reinterpret_cast<const i32x4 *>(smem + at);
#pragma unroll
		for (int q = 0; q < 4; ++q)
		{
			const i32x4 v = plane0[q * PLANE_ROWS];
			w[4 * q] = v.x;
			w[4 * q + 1] = v.y;
			w[4 * q + 2] = v.z;
			w[4 * q + 3] = v.w;
		}
	};
#pragma unroll
	for (int q = 0; q < 16; ++q)
		wa[q] = wb[q] = q + 1;

	auto one = [&](const int (&w)[16]) {
		int acc0, acc1;
		if constexpr (FORM == 0 || FORM == 1 || FORM == 4)
		{
			int lo[4], hi[4] = {0, 0, 0, 0};
#define TAP(K, LO, HI, SAMPLE, WEIGHT)                                                                                             \
	asm("v_ashrrev_i32_e32 v" #LO ", 31, %2\n\tv_mad_i64_i32 v[" #LO ":" #HI "], vcc, %2, %3, v[" #LO ":" #HI "]"                 \
	    : "=&{v" #LO "}"(lo[K]), "+{v" #HI "}"(hi[K]) : "v"(SAMPLE), "v"(WEIGHT) : "vcc")
#define TAPMOV(K, LO, HI, SAMPLE, WEIGHT, BIAS)                                                                                    \
	lo[K] = (BIAS);                                                                                                               \
	asm("v_mad_i64_i32 v[" #LO ":" #HI "], vcc, %2, %3, v[" #LO ":" #HI "]" : "+{v" #LO "}"(lo[K]), "+{v" #HI "}"(hi[K]) : "v"(SAMPLE), "v"(WEIGHT) : "vcc")
#pragma unroll
			for (int s = 0; s < TT; ++s)
			{
				if constexpr (FORM == 0)
				{
					TAP(0, 120, 121, S[s][0], w[s]);
					TAP(2, 124, 125, S[s][1], w[s]);
				}
				else if constexpr (FORM == 1)
				{
					if (s & 1)
					{
						TAP(1, 122, 123, S[s][0], w[s]);
						TAP(3, 126, 127, S[s][1], w[s]);
					}
					else
					{
						TAP(0, 120, 121, S[s][0], w[s]);
						TAP(2, 124, 125, S[s][1], w[s]);
					}
				}
				else
				{
					if (s & 1)
					{
						TAPMOV(1, 122, 123, S[s][0], w[s], B[s][0]);
						TAPMOV(3, 126, 127, S[s][1], w[s], B[s][1]);
					}
					else
					{
						TAPMOV(0, 120, 121, S[s][0], w[s], B[s][0]);
						TAPMOV(2, 124, 125, S[s][1], w[s], B[s][1]);
					}
				}
			}
			acc0 = hi[0] + hi[1];
			acc1 = hi[2] + hi[3];
		}
		else if constexpr (FORM == 2)
		{
			int h0 = 0, h1 = 0, l0, l1;
#define T2(s) "v_ashrrev_i32_e32 v120, 31, %[a" #s "]\n\tv_mad_i64_i32 v[120:121], vcc, %[a" #s "], %[w" #s "], v[120:121]\n\t"      \
              "v_ashrrev_i32_e32 v124, 31, %[b" #s "]\n\tv_mad_i64_i32 v[124:125], vcc, %[b" #s "], %[w" #s "], v[124:125]\n\t"
#define OPS(s) [a##s] "v"(S[s][0]), [b##s] "v"(S[s][1]), [w##s] "v"(w[s])
			asm(T2(0) T2(1) T2(2) T2(3) T2(4) T2(5) T2(6) T2(7) T2(8) T2(9) T2(10) T2(11) T2(12) T2(13) T2(14) "s_nop 0"
			    : "=&{v120}"(l0), "+{v121}"(h0), "=&{v124}"(l1), "+{v125}"(h1)
			    : OPS(0), OPS(1), OPS(2), OPS(3), OPS(4), OPS(5), OPS(6), OPS(7), OPS(8), OPS(9), OPS(10), OPS(11), OPS(12), OPS(13), OPS(14)
			    : "vcc");
			acc0 = h0;
			acc1 = h1;
		}
		else if constexpr (FORM == 3 || FORM == 6 || FORM == 7)
		{
			int h0 = 0, h1 = 0, h2 = 0, h3 = 0, l0, l1, l2, l3;
#define T3E(s) "v_ashrrev_i32_e32 v120, 31, %[a" #s "]\n\tv_mad_i64_i32 v[120:121], vcc, %[a" #s "], %[w" #s "], v[120:121]\n\t"     \
               "v_ashrrev_i32_e32 v124, 31, %[b" #s "]\n\tv_mad_i64_i32 v[124:125], vcc, %[b" #s "], %[w" #s "], v[124:125]\n\t"
#define T3O(s) "v_ashrrev_i32_e32 v122, 31, %[a" #s "]\n\tv_mad_i64_i32 v[122:123], vcc, %[a" #s "], %[w" #s "], v[122:123]\n\t"     \
               "v_ashrrev_i32_e32 v126, 31, %[b" #s "]\n\tv_mad_i64_i32 v[126:127], vcc, %[b" #s "], %[w" #s "], v[126:127]\n\t"
			asm(T3E(0) T3O(1) T3E(2) T3O(3) T3E(4) T3O(5) T3E(6) T3O(7) T3E(8) T3O(9) T3E(10) T3O(11) T3E(12) T3O(13) T3E(14) "s_nop 0"
			    : "=&{v120}"(l0), "+{v121}"(h0), "=&{v124}"(l1), "+{v125}"(h1), "=&{v122}"(l2), "+{v123}"(h2), "=&{v126}"(l3), "+{v127}"(h3)
			    : OPS(0), OPS(1), OPS(2), OPS(3), OPS(4), OPS(5), OPS(6), OPS(7), OPS(8), OPS(9), OPS(10), OPS(11), OPS(12), OPS(13), OPS(14)
			    : "vcc");
			acc0 = h0 + h2;
			acc1 = h1 + h3;
		}
		else if constexpr (FORM == 8 || FORM == 10 || FORM == 11)
		{
			// 4 pairs; the shift (or move) that arms a pair's low dword is issued well before the multiply-add that consumes it
			int h0 = 0, h1 = 0, h2 = 0, h3 = 0, l0, l1, l2, l3;
#define ARM_E(s) "v_ashrrev_i32_e32 v120, 31, %[a" #s "]\n\tv_ashrrev_i32_e32 v124, 31, %[b" #s "]\n\t"
#define ARM_O(s) "v_ashrrev_i32_e32 v122, 31, %[a" #s "]\n\tv_ashrrev_i32_e32 v126, 31, %[b" #s "]\n\t"
#define MAD_E(s) "v_mad_i64_i32 v[120:121], vcc, %[a" #s "], %[w" #s "], v[120:121]\n\tv_mad_i64_i32 v[124:125], vcc, %[b" #s "], %[w" #s "], v[124:125]\n\t"
#define MAD_O(s) "v_mad_i64_i32 v[122:123], vcc, %[a" #s "], %[w" #s "], v[122:123]\n\tv_mad_i64_i32 v[126:127], vcc, %[b" #s "], %[w" #s "], v[126:127]\n\t"
			if constexpr (FORM == 8)
			{
				// arm s+1 between the multiply-adds of s-1 and s
				asm(ARM_E(0) ARM_O(1) MAD_E(0) ARM_E(2) MAD_O(1) ARM_O(3) MAD_E(2) ARM_E(4) MAD_O(3) ARM_O(5) MAD_E(4) ARM_E(6) MAD_O(5) ARM_O(7) MAD_E(6) ARM_E(8)
				    MAD_O(7) ARM_O(9) MAD_E(8) ARM_E(10) MAD_O(9) ARM_O(11) MAD_E(10) ARM_E(12) MAD_O(11) ARM_O(13) MAD_E(12) ARM_E(14) MAD_O(13) MAD_E(14) "s_nop 0"
				    : "=&{v120}"(l0), "+{v121}"(h0), "=&{v124}"(l1), "+{v125}"(h1), "=&{v122}"(l2), "+{v123}"(h2), "=&{v126}"(l3), "+{v127}"(h3)
				    : OPS(0), OPS(1), OPS(2), OPS(3), OPS(4), OPS(5), OPS(6), OPS(7), OPS(8), OPS(9), OPS(10), OPS(11), OPS(12), OPS(13), OPS(14)
				    : "vcc");
			}
			else if constexpr (FORM == 10)
			{
				// arm both slots of a pair of slots, then their four multiply-adds
				asm(ARM_E(0) ARM_O(1) MAD_E(0) MAD_O(1) ARM_E(2) ARM_O(3) MAD_E(2) MAD_O(3) ARM_E(4) ARM_O(5) MAD_E(4) MAD_O(5) ARM_E(6) ARM_O(7) MAD_E(6) MAD_O(7)
				    ARM_E(8) ARM_O(9) MAD_E(8) MAD_O(9) ARM_E(10) ARM_O(11) MAD_E(10) MAD_O(11) ARM_E(12) ARM_O(13) MAD_E(12) MAD_O(13) ARM_E(14) MAD_E(14) "s_nop 0"
				    : "=&{v120}"(l0), "+{v121}"(h0), "=&{v124}"(l1), "+{v125}"(h1), "=&{v122}"(l2), "+{v123}"(h2), "=&{v126}"(l3), "+{v127}"(h3)
				    : OPS(0), OPS(1), OPS(2), OPS(3), OPS(4), OPS(5), OPS(6), OPS(7), OPS(8), OPS(9), OPS(10), OPS(11), OPS(12), OPS(13), OPS(14)
				    : "vcc");
			}
			else
			{
				// the multiply-adds only (low dwords never re-armed: WRONG arithmetic, timing only): what the arming costs
				asm(MAD_E(0) MAD_O(1) MAD_E(2) MAD_O(3) MAD_E(4) MAD_O(5) MAD_E(6) MAD_O(7) MAD_E(8) MAD_O(9) MAD_E(10) MAD_O(11) MAD_E(12) MAD_O(13) MAD_E(14) "s_nop 0"
				    : "=&{v120}"(l0), "+{v121}"(h0), "=&{v124}"(l1), "+{v125}"(h1), "=&{v122}"(l2), "+{v123}"(h2), "=&{v126}"(l3), "+{v127}"(h3)
				    : OPS(0), OPS(1), OPS(2), OPS(3), OPS(4), OPS(5), OPS(6), OPS(7), OPS(8), OPS(9), OPS(10), OPS(11), OPS(12), OPS(13), OPS(14)
				    : "vcc");
			}
			acc0 = h0 + h2;
			acc1 = h1 + h3;
		}
		else if constexpr (FORM == 9)
		{
			// 2 pairs: arm both channels, then both multiply-adds
			int h0 = 0, h1 = 0, l0, l1;
#define T9(s) "v_ashrrev_i32_e32 v120, 31, %[a" #s "]\n\tv_ashrrev_i32_e32 v124, 31, %[b" #s "]\n\t"                               \
              "v_mad_i64_i32 v[120:121], vcc, %[a" #s "], %[w" #s "], v[120:121]\n\tv_mad_i64_i32 v[124:125], vcc, %[b" #s "], %[w" #s "], v[124:125]\n\t"
			asm(T9(0) T9(1) T9(2) T9(3) T9(4) T9(5) T9(6) T9(7) T9(8) T9(9) T9(10) T9(11) T9(12) T9(13) T9(14) "s_nop 0"
			    : "=&{v120}"(l0), "+{v121}"(h0), "=&{v124}"(l1), "+{v125}"(h1)
			    : OPS(0), OPS(1), OPS(2), OPS(3), OPS(4), OPS(5), OPS(6), OPS(7), OPS(8), OPS(9), OPS(10), OPS(11), OPS(12), OPS(13), OPS(14)
			    : "vcc");
			acc0 = h0;
			acc1 = h1;
		}
		else if constexpr (FORM >= 14 && FORM <= 19)
		{
			// 2 pairs, one statement per frame (as form 2 / 12), the low dword armed by ...: which single instructions are as cheap
			// as form 12's v_mov_b32?  (14, 17: right arithmetic - bits 31..7 the sign, low bits noise the carry does not see;
			// 15, 18: right; 16, 19: timing only)
			int h0 = 0, h1 = 0, l0, l1;
#define ARM14(r, x) "v_mov_b32_sdwa v" #r ", sext(%[" #x "]) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3\n\t"
#define ARM15(r, x) "v_bfe_i32 v" #r ", %[" #x "], 31, 1\n\t"
#define ARM16(r, x) "v_not_b32_e32 v" #r ", %[" #x "]\n\t"
#define ARM17(r, x, w) "v_xor_b32_sdwa v" #r ", sext(%[" #x "]), sext(%[" #w "]) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3\n\t"
#define ARM18(r, x) "v_ashrrev_i32_e64 v" #r ", 31, %[" #x "]\n\t"
#define ARM19(r, x) "v_and_b32_e32 v" #r ", 0x7f, %[" #x "]\n\t"
#define MADS(s) "v_mad_i64_i32 v[120:121], vcc, %[a" #s "], %[w" #s "], v[120:121]\n\t"
#define MADT(s) "v_mad_i64_i32 v[124:125], vcc, %[b" #s "], %[w" #s "], v[124:125]\n\t"
#define FRAME(A, B) A(0) B(0) A(1) B(1) A(2) B(2) A(3) B(3) A(4) B(4) A(5) B(5) A(6) B(6) A(7) B(7) A(8) B(8) A(9) B(9) A(10) B(10) A(11) B(11) A(12) B(12) A(13) B(13) A(14) B(14)
#define OUTS : "=&{v120}"(l0), "+{v121}"(h0), "=&{v124}"(l1), "+{v125}"(h1)
#define INS : OPS(0), OPS(1), OPS(2), OPS(3), OPS(4), OPS(5), OPS(6), OPS(7), OPS(8), OPS(9), OPS(10), OPS(11), OPS(12), OPS(13), OPS(14) : "vcc"
#define L14(s) ARM14(120, a##s) MADS(s)
#define R14(s) ARM14(124, b##s) MADT(s)
#define L15(s) ARM15(120, a##s) MADS(s)
#define R15(s) ARM15(124, b##s) MADT(s)
#define L16(s) ARM16(120, a##s) MADS(s)
#define R16(s) ARM16(124, b##s) MADT(s)
#define L17(s) ARM17(120, a##s, w##s) MADS(s)
#define R17(s) ARM17(124, b##s, w##s) MADT(s)
#define L18(s) ARM18(120, a##s) MADS(s)
#define R18(s) ARM18(124, b##s) MADT(s)
#define L19(s) ARM19(120, a##s) MADS(s)
#define R19(s) ARM19(124, b##s) MADT(s)
			if constexpr (FORM == 14)
				asm(FRAME(L14, R14) "s_nop 0" OUTS INS);
			else if constexpr (FORM == 15)
				asm(FRAME(L15, R15) "s_nop 0" OUTS INS);
			else if constexpr (FORM == 16)
				asm(FRAME(L16, R16) "s_nop 0" OUTS INS);
			else if constexpr (FORM == 17)
				asm(FRAME(L17, R17) "s_nop 0" OUTS INS);
			else if constexpr (FORM == 18)
				asm(FRAME(L18, R18) "s_nop 0" OUTS INS);
			else
				asm(FRAME(L19, R19) "s_nop 0" OUTS INS);
			acc0 = h0;
			acc1 = h1;
		}
		else if constexpr (FORM == 12 || FORM == 13)
		{
			// 2 pairs (one per channel), low dword armed by a MOVE from a bias register (B = S >> 31, formed once per window)
			int h0 = 0, h1 = 0, l0, l1;
			if constexpr (FORM == 12)
			{
#define OPSB(s) [a##s] "v"(S[s][0]), [b##s] "v"(S[s][1]), [w##s] "v"(w[s]), [p##s] "v"(B[s][0]), [q##s] "v"(B[s][1])
#define T12(s) "v_mov_b32_e32 v120, %[p" #s "]\n\tv_mov_b32_e32 v124, %[q" #s "]\n\t"                                               \
               "v_mad_i64_i32 v[120:121], vcc, %[a" #s "], %[w" #s "], v[120:121]\n\tv_mad_i64_i32 v[124:125], vcc, %[b" #s "], %[w" #s "], v[124:125]\n\t"
				asm(T12(0) T12(1) T12(2) T12(3) T12(4) T12(5) T12(6) T12(7) T12(8) T12(9) T12(10) T12(11) T12(12) T12(13) T12(14) "s_nop 0"
				    : "=&{v120}"(l0), "+{v121}"(h0), "=&{v124}"(l1), "+{v125}"(h1)
				    : OPSB(0), OPSB(1), OPSB(2), OPSB(3), OPSB(4), OPSB(5), OPSB(6), OPSB(7), OPSB(8), OPSB(9), OPSB(10), OPSB(11), OPSB(12), OPSB(13), OPSB(14)
				    : "vcc");
			}
			else
			{
#define TAPM(LO, HI, VLO, VHI, SAMPLE, WEIGHT, BIAS)                                                                               \
	VLO = (BIAS);                                                                                                                 \
	asm("v_mad_i64_i32 v[" #LO ":" #HI "], vcc, %2, %3, v[" #LO ":" #HI "]" : "+{v" #LO "}"(VLO), "+{v" #HI "}"(VHI) : "v"(SAMPLE), "v"(WEIGHT) : "vcc")
#pragma unroll
				for (int s = 0; s < TT; ++s)
				{
					TAPM(120, 121, l0, h0, S[s][0], w[s], B[s][0]);
					TAPM(124, 125, l1, h1, S[s][1], w[s], B[s][1]);
				}
			}
			acc0 = h0;
			acc1 = h1;
		}
		else
		{
			acc0 = (__mul24(S[0][0], w[0]) + B[0][0]) >> 16;
			acc1 = (__mul24(S[0][1], w[0]) + B[0][1]) >> 16;
#pragma unroll
			for (int s = 1; s < TT; ++s)
			{
				int x0, x1;
				asm("v_mad_i32_i24 %2, %4, %6, %7\n\t"
				    "v_mad_i32_i24 %3, %5, %6, %8\n\t"
				    "v_add_u32_sdwa %0, %0, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
				    "v_add_u32_sdwa %1, %1, sext(%3) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
				    : "+v"(acc0), "+v"(acc1), "=&v"(x0), "=&v"(x1)
				    : "v"(S[s][0]), "v"(S[s][1]), "v"(w[s]), "v"(B[s][0]), "v"(B[s][1]));
			}
		}
		const long long v0 = (long long)acc0 * (long long)w[TT] + (long long)((unsigned)(acc0 >> 31) >> 17);
		const long long v1 = (long long)acc1 * (long long)w[TT] + (long long)((unsigned)(acc1 >> 31) >> 17);
		i32x2 q;
		q.x = (int)(v0 >> 15);
		q.y = (int)(v1 >> 15);
		if constexpr (FORM == 7)
			asm volatile("" ::"v"(q.x), "v"(q.y));
		else
			*reinterpret_cast<i32x2 *>(stage + (stage_at & 0xFF8u)) = q;
	};

	read_row(g, wa);
	for (int j = 0; j < frames; j += 2)
	{
		read_row(g - increment, wb);
		__builtin_amdgcn_sched_barrier(0);
		one(wa);
		stage_at += 8u;
		read_row(g - 2u * increment, wa);
		__builtin_amdgcn_sched_barrier(0);
		one(wb);
		stage_at += 8u;
		g -= 2u * increment;
	}
	if (blockIdx.x == 0 && tid == 0)
	{
		stamps[0] = __builtin_amdgcn_s_memtime() - c0;
		stamps[1] = __builtin_amdgcn_s_memrealtime() - r0;
	}
	if (stage[lane * 8u] == 0x5A && wa[3] == 0x12345678)
		out[tid] = wb[2];
}

template <int FORM, int WAVES>
static void run(const char *name, int *d)
{
	const int frames = 4000, blocks = 256;
	const unsigned lds = 4u * PLANE_ROWS * 16u + WAVES * (WAVES > 12 ? 4096u : 6144u) + (WAVES == 12 ? 20000u : 0u);   // 12 waves: pad so that one workgroup fills the CU
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	unsigned long long *stamps;
	CHECK(hipMalloc(&stamps, 16));
	CHECK(hipFuncSetAttribute((const void *)k<FORM, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	for (int w = 0; w < 20; ++w)
		hipLaunchKernelGGL((k<FORM, WAVES>), dim3(blocks), dim3(WAVES * 64), lds, 0, d, frames, 5461u, stamps);
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(e0));
	hipLaunchKernelGGL((k<FORM, WAVES>), dim3(blocks), dim3(WAVES * 64), lds, 0, d, frames, 5461u, stamps);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	unsigned long long h[2];
	CHECK(hipMemcpy(h, stamps, 16, hipMemcpyDeviceToHost));
	const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
	const double wave_frames_per_simd = (double)frames * WAVES / 4.0;
	printf("form %2d %-58s %2d waves/CU  %7.3f ms  clock %.2f GHz  %6.1f cycles per wave-frame per SIMD (whole launch), %6.1f (wave 0 of workgroup 0 alone)\n", FORM, name, WAVES, ms, ghz,
	       ms * 1e-3 * ghz * 1e9 / wave_frames_per_simd, (double)h[0] / frames);
}

int main()
{
	int *d;
	CHECK(hipMalloc(&d, 1 << 16));
	run<0, 12>("ashr + mad64, 2 pairs, statement per tap", d);
	run<1, 12>("ashr + mad64, 4 pairs, statement per tap", d);
	run<2, 12>("ashr + mad64, 2 pairs, one statement per frame", d);
	run<3, 12>("ashr + mad64, 4 pairs, one statement per frame", d);
	run<4, 12>("mov bias + mad64, 4 pairs, statement per tap (round 1)", d);
	run<5, 12>("mad_i32_i24 + add_sdwa", d);
	run<6, 12>("as 3, weights in registers (no LDS row reads)", d);
	run<7, 12>("as 3, no staging store", d);
	run<8, 12>("4 pairs, armed one slot ahead, one statement", d);
	run<9, 12>("2 pairs, arm L R then mad L R, one statement", d);
	run<10, 12>("4 pairs, arm 2 slots then 4 mads, one statement", d);
	run<11, 12>("4 pairs, multiply-adds only (timing only)", d);
	run<12, 12>("mov bias + mad64, 2 pairs, one statement per frame", d);
	run<13, 12>("mov bias + mad64, 2 pairs, statement per tap", d);
	run<2, 12>("ashr + mad64, 2 pairs, one statement per frame (again)", d);
	run<12, 12>("mov bias + mad64, 2 pairs, one statement per frame (again)", d);
	run<14, 12>("v_mov_b32_sdwa sext(byte 3) + mad64", d);
	run<15, 12>("v_bfe_i32 + mad64", d);
	run<16, 12>("v_not_b32 + mad64 (timing only)", d);
	run<17, 12>("v_xor_b32_sdwa sext(byte 3), sext(byte 3) + mad64", d);
	run<18, 12>("v_ashrrev_i32_e64 + mad64", d);
	run<19, 12>("v_and_b32 + mad64 (timing only)", d);
	run<14, 16>("v_mov_b32_sdwa sext(byte 3) + mad64", d);
	run<2, 16>("ashr + mad64, 2 pairs, one statement per frame", d);
	run<13, 16>("mov bias + mad64, 2 pairs, statement per tap", d);
	run<8, 16>("4 pairs, armed one slot ahead, one statement", d);
	run<4, 16>("mov bias + mad64, 4 pairs, statement per tap (round 1)", d);
	run<8, 8>("4 pairs, armed one slot ahead, one statement", d);
	return 0;
}
