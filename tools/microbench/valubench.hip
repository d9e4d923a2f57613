// valubench.hip - issue rate of the integer VALU instructions the tap arithmetic is made of, on MI355X.
// Every wave runs ITER x 64 independent instructions of one kind (8 accumulators round-robin); 8 waves per SIMD.
// Prints wave-instructions per cycle per SIMD (from the measured time and an assumed 2.4 GHz) and Tlane-ops/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int KIND>
__global__ __launch_bounds__(512) void k(int *out, int iters, int seed, unsigned long long *stamps)
{
	const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	int a[8];
#pragma unroll
	for (int i = 0; i < 8; ++i)
		a[i] = seed + threadIdx.x * (i + 1);
	int w = seed * 3 + 1;
	long long b[8];
#pragma unroll
	for (int i = 0; i < 8; ++i)
		b[i] = a[i];
	for (int it = 0; it < iters; ++it)
	{
#define ADD(i) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
#define MUL(i) asm volatile("v_mul_i32_i24_e32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
#define ASHR(i) asm volatile("v_ashrrev_i32_e32 %0, 3, %0" : "+v"(a[i]));
#define ADDSDWA(i) asm volatile("v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a[i]) : "v"(w));
#define MULSDWA(i) asm volatile("v_mul_i32_i24_sdwa %0, sext(%0), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(w));
#define MAD(i) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(w));
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(w));
#define ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(w));
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
#define BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(w));
#define AND(i) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(w));
#define LSHR(i) asm volatile("v_lshrrev_b32_e32 %0, 1, %0" : "+v"(a[i]));
#define PKADD(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(w));
#define PKMUL(i) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(w));
#define PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(w));
#define MULU(i) asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
#define CNDM(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(w));
#define MADU16(i) asm volatile("v_mad_u32_u16 %0, %0, %1, %1" : "+v"(a[i]) : "v"(w));
#define ADDC(i) asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(w) : "vcc");
#define MULHI(i) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
#define MULHI24(i) asm volatile("v_mul_hi_i32_i24_e32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
#define MAD64(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %1, %0" : "+v"(b[i]) : "v"(w) : "vcc");
#define DOT2(i) asm volatile("v_dot2_i32_i16 %0, %0, %1, %1" : "+v"(a[i]) : "v"(w));
#define XAD(i) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(w));
#define LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(w));
		if constexpr (KIND == 0) { REP64(ADD) }
		if constexpr (KIND == 1) { REP64(MUL) }
		if constexpr (KIND == 2) { REP64(ASHR) }
		if constexpr (KIND == 3) { REP64(ADDSDWA) }
		if constexpr (KIND == 4) { REP64(MULSDWA) }
		if constexpr (KIND == 5) { REP64(MAD) }
		if constexpr (KIND == 6) { REP64(FMA) }
		if constexpr (KIND == 7) { REP64(ADD3) }
		if constexpr (KIND == 8) { REP64(MULLO) }
		if constexpr (KIND == 9) { REP64(BFI) }
		if constexpr (KIND == 10) { REP64(AND) }
		if constexpr (KIND == 11) { REP64(LSHR) }
		if constexpr (KIND == 12) { REP64(PKADD) }
		if constexpr (KIND == 13) { REP64(PKMUL) }
		if constexpr (KIND == 14) { REP64(PERM) }
		if constexpr (KIND == 15) { REP64(MULU) }
		if constexpr (KIND == 16) { REP64(CNDM) }
		if constexpr (KIND == 17) { REP64(MADU16) }
		if constexpr (KIND == 18) { REP64(ADDC) }
		if constexpr (KIND == 19) { REP64(MULHI) }
		if constexpr (KIND == 20) { REP64(MULHI24) }
		if constexpr (KIND == 21) { REP64(MAD64) }
		if constexpr (KIND == 22) { REP64(DOT2) }
		if constexpr (KIND == 23) { REP64(XAD) }
		if constexpr (KIND == 24) { REP64(LSHLADD) }
	}
	if (blockIdx.x == 0 && threadIdx.x == 0)
	{
		stamps[0] = __builtin_amdgcn_s_memtime() - c0;
		stamps[1] = __builtin_amdgcn_s_memrealtime() - r0;
	}
	int r = 0;
#pragma unroll
	for (int i = 0; i < 8; ++i)
		r ^= a[i] ^ (int)b[i] ^ (int)(b[i] >> 32);
	if (r == 0x12345678)
		out[threadIdx.x] = r;
}

template <int KIND>
static void run(const char *name, int *d)
{
	const int iters = 2000, blocks = 256 * 4; // 4 blocks of 512 threads per CU = 32 waves/CU = 8 per SIMD
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	unsigned long long *stamps;
	CHECK(hipMalloc(&stamps, 16));
	for (int w = 0; w < 50; ++w)
		hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(512), 0, 0, d, iters, 7, stamps);
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(e0));
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(512), 0, 0, d, iters, 7, stamps);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	unsigned long long h[2];
	CHECK(hipMemcpy(h, stamps, 16, hipMemcpyDeviceToHost));
	const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
	const double wave_instr = (double)blocks * 8 * iters * 64;
	const double per_simd_per_s = wave_instr / 1024.0 / (ms * 1e-3);
	printf("%-22s %8.3f ms  in-kernel clock %.2f GHz  %.2f cycles per wave-instr per SIMD at that clock  %.1f Tlane-op/s\n", name, ms, ghz, ghz * 1e9 / per_simd_per_s, wave_instr * 64 / (ms * 1e-3) / 1e12);
}

int main()
{
	int *d;
	CHECK(hipMalloc(&d, 4096));
	run<0>("v_add_u32", d);
	run<1>("v_mul_i32_i24", d);
	run<2>("v_ashrrev_i32", d);
	run<3>("v_add_u32_sdwa", d);
	run<4>("v_mul_i32_i24_sdwa", d);
	run<5>("v_mad_i32_i24", d);
	run<6>("v_fma_f32", d);
	run<7>("v_add3_u32", d);
	run<8>("v_mul_lo_u32", d);
	run<9>("v_bfi_b32", d);
	run<10>("v_and_b32", d);
	run<11>("v_lshrrev_b32", d);
	run<12>("v_pk_add_u16", d);
	run<13>("v_pk_mul_lo_u16", d);
	run<14>("v_perm_b32", d);
	run<15>("v_mul_u32_u24", d);
	run<16>("v_cndmask_b32", d);
	run<17>("v_mad_u32_u16", d);
	run<18>("v_add_co_u32", d);
	run<19>("v_mul_hi_i32", d);
	run<20>("v_mul_hi_i32_i24", d);
	run<21>("v_mad_i64_i32", d);
	run<22>("v_dot2_i32_i16", d);
	run<23>("v_xad_u32", d);
	run<24>("v_lshl_add_u32", d);
	return 0;
}
