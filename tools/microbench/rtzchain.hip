// rtzchain.hip - can a chain of FP32 fused multiply-adds under round-toward-zero stand in for the per-tap truncation
// (clownresampler.h:1020: acc += (sample * weight) / 65536, C division)?  An accumulator that starts at +2^23 has an ulp of exactly 1;
// fma(|s|, |w| / 65536, acc) under RTZ is acc + floor(|s| |w| / 65536), the product exact inside the fused operation.  Products
// that are negative go to a second chain that starts at -2^23 (RTZ on a negative sum truncates toward zero too); both chains
// advance in ONE v_pk_fma_f32.  Checks 2^24 random 15-tap frames per launch against the integer definition, edge samples and
// weights included, and times the chain against v_mov_b32 + v_mad_i64_i32 (the form k_up2 used).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int TT = 15;

__device__ __forceinline__ unsigned rnd(unsigned &x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }

__global__ void check(unsigned seed, unsigned long long *bad, int *first_bad)
{
	unsigned x = seed + 0x9E3779B9u * (blockIdx.x * blockDim.x + threadIdx.x + 1u);
	int s[TT], w[TT];
	for (int k = 0; k < TT; ++k)
	{
		const unsigned r = rnd(x);
		s[k] = (int)(short)(r & 0xFFFF);
		w[k] = (int)((r >> 16) % 65537u);                    // 0 .. 65536
		const unsigned e = rnd(x) & 63u;
		if (e == 0) s[k] = -32768; else if (e == 1) s[k] = 32767; else if (e == 2) s[k] = 0; else if (e == 3) w[k] = 65536; else if (e == 4) w[k] = 0; else if (e == 5) w[k] = 1;
		if (rnd(x) & 1u) w[k] = -w[k] == -65536 ? -65535 : -w[k];   // (the kernels take -65536 < weight <= 65536)
	}
	long long want = 0;
	for (int k = 0; k < TT; ++k)
		want += ((long long)s[k] * w[k]) / 65536;            // C division: toward zero
	const f32x2 base = {8388608.0f, -8388608.0f};
	f32x2 acc = base;
	asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3");
	for (int k = 0; k < TT; ++k)
	{
		const int v = w[k] < 0 ? -s[k] : s[k];               // the weight's sign folded into the sample
		const float vf = (float)v;
		f32x2 sp;
		sp.x = vf > 0.0f ? vf : 0.0f;
		sp.y = vf < 0.0f ? vf : 0.0f;
		f32x2 wp;
		wp.x = (float)(w[k] < 0 ? -w[k] : w[k]) * (1.0f / 65536.0f);
		wp.y = 0.0f;
		asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(sp), "v"(wp));
	}
	asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0");
	const int got = (int)(__float_as_uint(acc.x) - __float_as_uint(acc.y) + 0x80000000u);
	if ((long long)got != want)
	{
		if (atomicAdd(bad, 1ull) == 0)
		{
			first_bad[0] = got;
			first_bad[1] = (int)want;
		}
	}
}

// throughput: FRAMES frames of 15 taps x 2 channels per lane; A = pk_fma chain, B = mov + mad_i64 chain
template <int FORM>
__global__ __launch_bounds__(768) void rate(const int *in, int *out, int frames)
{
	int f[TT];
	for (int k = 0; k < TT; ++k)
		f[k] = in[(threadIdx.x + k) & 1023];
	int total = 0;
	if constexpr (FORM == 0)
	{
		f32x2 S[TT][2];
		for (int k = 0; k < TT; ++k)
			for (int c = 0; c < 2; ++c)
			{
				const float vf = (float)(c ? (f[k] >> 16) : (int)(short)f[k]);
				S[k][c].x = vf > 0.0f ? vf : 0.0f;
				S[k][c].y = vf < 0.0f ? vf : 0.0f;
				asm volatile("" : "+v"(S[k][c]));
			}
		const f32x2 base = {8388608.0f, -8388608.0f};
		asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3");
		for (int j = 0; j < frames; ++j)
		{
			f32x2 w[8];
			for (int k = 0; k < 8; ++k)
			{
				w[k].x = (float)((j + k) & 1023) * (1.0f / 65536.0f);
				w[k].y = w[k].x + 0.25f;
				asm volatile("" : "+v"(w[k]));
			}
			f32x2 a0 = base, a1 = base;
#pragma unroll
			for (int k = 0; k < TT; ++k)
			{
				if (k & 1)
				{
					asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a0) : "v"(S[k][0]), "v"(w[k / 2]));
					asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a1) : "v"(S[k][1]), "v"(w[k / 2]));
				}
				else
				{
					asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a0) : "v"(S[k][0]), "v"(w[k / 2]));
					asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a1) : "v"(S[k][1]), "v"(w[k / 2]));
				}
			}
			total += (int)(__float_as_uint(a0.x) - __float_as_uint(a0.y)) + (int)(__float_as_uint(a1.x) - __float_as_uint(a1.y));
		}
		asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0");
	}
	else
	{
		int X[TT][2];
		for (int k = 0; k < TT; ++k)
		{
			X[k][0] = 2 * (int)(short)f[k];
			X[k][1] = 2 * (f[k] >> 16);
			asm volatile("" : "+v"(X[k][0]), "+v"(X[k][1]));
		}
		for (int j = 0; j < frames; ++j)
		{
			int w[TT];
			for (int k = 0; k < TT; ++k)
			{
				w[k] = ((j + k) & 1023) << 15;
				asm volatile("" : "+v"(w[k]));
			}
			int lo0, hi0 = 0, lo1, hi1 = 0;
#pragma unroll
			for (int k = 0; k < TT; ++k)
			{
				lo0 = X[k][0];
				asm volatile("v_mad_i64_i32 v[120:121], vcc, %1, %2, v[120:121]" : "+{v121}"(hi0) : "v"(X[k][0]), "v"(w[k]), "{v120}"(lo0) : "vcc");
				lo1 = X[k][1];
				asm volatile("v_mad_i64_i32 v[124:125], s[94:95], %1, %2, v[124:125]" : "+{v125}"(hi1) : "v"(X[k][1]), "v"(w[k]), "{v124}"(lo1) : "s94", "s95");
			}
			total += hi0 + hi1;
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = total;
}

int main()
{
	unsigned long long *d_bad;
	int *d_first;
	CHECK(hipMalloc(&d_bad, 8));
	CHECK(hipMalloc(&d_first, 8));
	CHECK(hipMemset(d_bad, 0, 8));
	unsigned long long total_bad = 0;
	for (unsigned rep = 0; rep < 16; ++rep)
	{
		check<<<65536, 256>>>(12345u + 7919u * rep, d_bad, d_first);
		CHECK(hipDeviceSynchronize());
	}
	int first[2];
	CHECK(hipMemcpy(&total_bad, d_bad, 8, hipMemcpyDeviceToHost));
	CHECK(hipMemcpy(first, d_first, 8, hipMemcpyDeviceToHost));
	printf("RTZ pk_fma chain against sum of trunc(sample * weight / 65536): %llu mismatches in %llu frames of 15 taps", total_bad, 16ull * 65536 * 256);
	if (total_bad)
		printf(" (first: got %d want %d)", first[0], first[1]);
	printf("\n");

	int *d_in, *d_out;
	CHECK(hipMalloc(&d_in, 4096));
	CHECK(hipMalloc(&d_out, 256 * 768 * 4));
	std::vector<int> h(1024);
	for (int i = 0; i < 1024; ++i)
		h[i] = rand() * 65537;
	CHECK(hipMemcpy(d_in, h.data(), 4096, hipMemcpyHostToDevice));
	const int frames = 4096;
	for (int form = 0; form < 2; ++form)
		for (int rep = 0; rep < 3; ++rep)
		{
			hipEvent_t e0, e1;
			CHECK(hipEventCreate(&e0));
			CHECK(hipEventCreate(&e1));
			CHECK(hipEventRecord(e0));
			if (form == 0)
				rate<0><<<256, 768>>>(d_in, d_out, frames);
			else
				rate<1><<<256, 768>>>(d_in, d_out, frames);
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms;
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			// 256 workgroups of 12 waves = 3 waves per SIMD, like k_up2
			printf("%s: %d frames x 30 taps per lane, 3 waves per SIMD: %.3f ms = %.1f ns per wave-frame per SIMD\n", form == 0 ? "v_pk_fma_f32 (RTZ) chain " : "v_mov + v_mad_i64_i32   ",
			       frames, ms, ms * 1e6 / (frames * 3.0));
		}
	return 0;
}
