// chainlat.hip - what a DEPENDENT v_pk_fma_f32 costs: N independent accumulator chains interleaved (N = 1, 2, 4, 8), every instruction
// adding to the result of the one N instructions before it, at 1 .. 4 waves per SIMD.  ns per wave-instruction and SIMD (256 CUs busy).
// (valurate.hip prices the instruction with eight independent destinations; k_up2 / k_up3 run two chains per frame.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int N, int MOV>
__global__ void k(float *out, int reps, float seed)
{
	f32x2 acc[8], p, w;
	p.x = seed; p.y = -seed; w.x = 0.25f; w.y = 0.5f;
	for (int i = 0; i < 8; ++i) { acc[i].x = (float)i; acc[i].y = -(float)i; }
	for (int r = 0; r < reps; ++r)
	{
#pragma unroll
		for (int u = 0; u < 32; ++u)
		{
			if (MOV)
				asm volatile("v_mov_b32 %0, %1" : "+v"(acc[(u + 1) % N].x) : "v"(p.x));   // (an unrelated 1-pass VALU instruction in between)
			asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[u % N]) : "v"(p), "v"(w));
		}
	}
	float s = 0;
	for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int N, int MOV>
void run(int waves_per_simd)
{
	const int blocks = 256, threads = waves_per_simd * 4 * 64, reps = 2000;
	float *d; CHECK(hipMalloc(&d, blocks * threads * sizeof(float)));
	hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
	for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<N, MOV>), dim3(blocks), dim3(threads), 0, 0, d, reps, 1.5f);
	CHECK(hipEventRecord(a)); 
	for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<N, MOV>), dim3(blocks), dim3(threads), 0, 0, d, reps, 1.5f);
	CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
	float ms; CHECK(hipEventElapsedTime(&ms, a, b));
	const double inst = 5.0 * reps * 32.0 * (MOV ? 2 : 1) * waves_per_simd;   // wave-instructions per SIMD
	printf("chains %d%s, %d waves/SIMD: %.2f ns per wave-instruction and SIMD\n", N, MOV ? " (+ a v_mov between)" : "", waves_per_simd, ms * 1e6 / inst);
	CHECK(hipFree(d));
}

int main()
{
	for (int w = 1; w <= 4; ++w) { run<1, 0>(w); run<2, 0>(w); run<4, 0>(w); run<8, 0>(w); }
	for (int w = 3; w <= 4; ++w) { run<2, 1>(w); run<4, 1>(w); }
	return 0;
}
