// valurate.hip - issue cost of the VALU instructions the tap arithmetic can be built from, at k_up2's occupancy (3 waves per SIMD, all
// 1,024 SIMDs busy): ns and shader cycles (s_memtime) per wave-instruction per SIMD.  Eight independent destination registers per
// instruction class, so dependent-issue latency is not what is measured.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int KIND>
__global__ __launch_bounds__(768) void k(int *out, int loops, unsigned long long *cycles)
{
	int a[8], b = threadIdx.x | 1, c = threadIdx.x * 3 + 7;
	long long p[8];
	float f[8], g = 1.0f + threadIdx.x * 1e-6f;
	typedef float f32x2 __attribute__((ext_vector_type(2)));
	f32x2 q[8], h = {g, -g};
	for (int i = 0; i < 8; ++i) { a[i] = i + threadIdx.x; p[i] = i; f[i] = (float)i; q[i].x = (float)i; q[i].y = -(float)i; }
	asm volatile("" : "+v"(b), "+v"(c), "+v"(g), "+v"(h));
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int l = 0; l < loops; ++l)
	{
#define I_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
#define I_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define I_XAD(i) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define I_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define I_MUL24(i) asm volatile("v_mul_i32_i24 %0, %1, %2" : "=v"(a[i]) : "v"(b), "v"(c));
#define I_MAD24(i) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#define I_MULHI24(i) asm volatile("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(a[i]) : "v"(b), "v"(c));
#define I_MULHI(i) asm volatile("v_mul_hi_i32 %0, %1, %2" : "=v"(a[i]) : "v"(b), "v"(c));
#define I_SDWAADD(i) asm volatile("v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a[i]) : "v"(b));
#define I_SDWAMUL(i) asm volatile("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(a[i]) : "v"(b), "v"(c));
#define I_MAD64(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(p[i]) : "v"(b), "v"(c) : "vcc");
#define I_FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(g), "v"(g));
#define I_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(q[i]) : "v"(h), "v"(h));
#define I_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a[i]) : "v"(b));
#define I_CVT(i) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f[i]) : "v"(b));
#define I_MOVMAD(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b)); asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(p[i]) : "v"(b), "v"(c) : "vcc");
#define I_FMA2(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(g), "v"(g)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(q[i].x) : "v"(g), "v"(g));
// round 5 (k_seg): the weight as a SCALAR pair, the neg / swap modifiers of the negative slots, and the frame's real dependency shape -
// TWO accumulators taking turns (a chain per channel), with vector and with scalar weights
#define I_PKFMA_S(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(q[i]) : "v"(h), "s"(hs));
#define I_PKFMA_N(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(q[i]) : "v"(h), "v"(h));
#define I_PKFMA_NS(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(q[i]) : "v"(h), "s"(hs));
#define I_PKFMA_2(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(q[i & 1]) : "v"(h), "v"(h));
#define I_PKFMA_2S(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(q[i & 1]) : "v"(h), "s"(hs));
#define I_PKFMA_1(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(q[0]) : "v"(h), "v"(h));
		f32x2 hs;
		hs.x = __builtin_amdgcn_readfirstlane(__float_as_int(1.0f + loops * 1e-6f)) * 1e-9f;
		hs.y = hs.x + 1.0f;
		if constexpr (KIND == 17) { REP64(I_PKFMA_S) }
		if constexpr (KIND == 18) { REP64(I_PKFMA_N) }
		if constexpr (KIND == 19) { REP64(I_PKFMA_NS) }
		if constexpr (KIND == 20) { REP64(I_PKFMA_2) }
		if constexpr (KIND == 21) { REP64(I_PKFMA_2S) }
		if constexpr (KIND == 22) { REP64(I_PKFMA_1) }
		if constexpr (KIND == 0) { REP64(I_MOV) }
		if constexpr (KIND == 1) { REP64(I_ADD) }
		if constexpr (KIND == 2) { REP64(I_XAD) }
		if constexpr (KIND == 3) { REP64(I_ADD3) }
		if constexpr (KIND == 4) { REP64(I_MUL24) }
		if constexpr (KIND == 5) { REP64(I_MAD24) }
		if constexpr (KIND == 6) { REP64(I_MULHI24) }
		if constexpr (KIND == 7) { REP64(I_MULHI) }
		if constexpr (KIND == 8) { REP64(I_SDWAADD) }
		if constexpr (KIND == 9) { REP64(I_SDWAMUL) }
		if constexpr (KIND == 10) { REP64(I_MAD64) }
		if constexpr (KIND == 11) { REP64(I_FMA) }
		if constexpr (KIND == 12) { REP64(I_PKFMA) }
		if constexpr (KIND == 13) { REP64(I_ALIGNBIT) }
		if constexpr (KIND == 14) { REP64(I_CVT) }
		if constexpr (KIND == 15) { REP64(I_MOVMAD) }
		if constexpr (KIND == 16) { REP64(I_FMA2) }
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	int s = 0;
	for (int i = 0; i < 8; ++i) s += a[i] + (int)p[i] + (int)f[i] + (int)q[i].x + (int)q[i].y;
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int KIND>
void run(const char *name, int *d_out, unsigned long long *d_cyc)
{
	const int loops = 2000;
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	float best = 1e9f;
	unsigned long long cyc = 0;
	for (int rep = 0; rep < 3; ++rep)
	{
		CHECK(hipEventRecord(e0));
		k<KIND><<<256, 768>>>(d_out, loops, d_cyc);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) { best = ms; CHECK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost)); }
	}
	const double n = (double)loops * 64 * 3;   // wave-instructions (or pairs) per SIMD
	printf("%-44s %7.3f ns  %6.2f shader cycles per wave-instruction per SIMD (3 waves per SIMD)\n", name, best * 1e6 / n, (double)cyc / n);
}

int main()
{
	int *d_out;
	unsigned long long *d_cyc;
	CHECK(hipMalloc(&d_out, 256 * 768 * 4));
	CHECK(hipMalloc(&d_cyc, 8));
	run<0>("v_mov_b32", d_out, d_cyc);
	run<1>("v_add_u32", d_out, d_cyc);
	run<2>("v_xad_u32", d_out, d_cyc);
	run<3>("v_add3_u32", d_out, d_cyc);
	run<4>("v_mul_i32_i24", d_out, d_cyc);
	run<5>("v_mad_i32_i24", d_out, d_cyc);
	run<6>("v_mul_hi_u32_u24", d_out, d_cyc);
	run<7>("v_mul_hi_i32", d_out, d_cyc);
	run<8>("v_add_u32_sdwa (WORD_1 sext)", d_out, d_cyc);
	run<9>("v_mul_i32_i24_sdwa", d_out, d_cyc);
	run<10>("v_mad_i64_i32", d_out, d_cyc);
	run<11>("v_fma_f32", d_out, d_cyc);
	run<12>("v_pk_fma_f32", d_out, d_cyc);
	run<13>("v_alignbit_b32", d_out, d_cyc);
	run<14>("v_cvt_f32_i32_sdwa", d_out, d_cyc);
	run<15>("v_mov_b32 + v_mad_i64_i32 (per PAIR)", d_out, d_cyc);
	run<16>("v_fma_f32 + v_fma_f32 (per PAIR)", d_out, d_cyc);
	run<17>("v_pk_fma_f32, scalar weight pair", d_out, d_cyc);
	run<18>("v_pk_fma_f32, swapped + negated sample pair", d_out, d_cyc);
	run<19>("v_pk_fma_f32, swapped + negated, scalar weight", d_out, d_cyc);
	run<20>("v_pk_fma_f32, TWO accumulators taking turns", d_out, d_cyc);
	run<21>("v_pk_fma_f32, two accumulators, scalar weight", d_out, d_cyc);
	run<22>("v_pk_fma_f32, ONE accumulator (dependent chain)", d_out, d_cyc);
	return 0;
}
