// ldsvalu.hip - do a wave's LDS reads and its VALU work overlap, or add up?  16 waves per CU (4 per SIMD), each in a loop of
// K conflict-free ds_read_b64 (requested one trip ahead: no trip waits for its own reads) and M arithmetic instructions on
// registers that do not depend on the reads.  Reported: ns per trip per CU-wave for reads alone, arithmetic alone and both.
//   hipcc --offload-arch=gfx950 -O2 -o ldsvalu ldsvalu.hip && ./ldsvalu
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int WAVES = 16;

// OP: 0 v_mad_i64_i32 (+ arming v_mov) on 4 accumulator pairs, 1 v_pk_fma_f32 on 4 pairs, 2 v_add_u32 on 8 registers
template <int K, int M, int OP, int DEP>
__global__ __launch_bounds__(WAVES * 64) void k(unsigned trips, unsigned *out)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	unsigned *mine = reinterpret_cast<unsigned *>(smem + wave * 8192u);
	for (unsigned i = lane; i < 2048u; i += 64u)
		mine[i] = i * 2654435761u;
	__syncthreads();
	const unsigned at = (unsigned)(uintptr_t)mine + lane * 8u;
	i32x2 v[K > 0 ? K : 1];
	long long acc[4] = {1, 2, 3, 4};
	f32x2 facc[4] = {{1.f, 2.f}, {3.f, 4.f}, {5.f, 6.f}, {7.f, 8.f}};
	unsigned iacc[8] = {1, 2, 3, 4, 5, 6, 7, 8};
	int x = (int)(tid * 77u + 5u), wgt = (int)(tid * 13u + 3u);
	f32x2 fx = {1.0f + tid, 2.0f}, fw = {0.5f, 0.25f};
	asm volatile("" : "+v"(x), "+v"(wgt), "+v"(fx), "+v"(fw));
#pragma unroll
	for (int r = 0; r < K; ++r)
		asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[r]) : "v"(at), "n"(r * 512));
	unsigned sink = 0;
	for (unsigned t = 0; t < trips; ++t)
	{
		if constexpr (K > 0)
		{
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int r = 0; r < K; ++r)
			{
				asm volatile("" : "+v"(v[r]));
				if (DEP)
					x ^= v[r].x;       // the arithmetic uses what was read (one v_xor per read)
				else if (r == 0)
					sink ^= (unsigned)v[0].x;
			}
#pragma unroll
			for (int r = 0; r < K; ++r)
				asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[r]) : "v"(at), "n"(r * 512));
		}
#pragma unroll
		for (int m = 0; m < M; ++m)
		{
			if constexpr (OP == 0)
			{
				int lo = x;   // (the arming move)
				asm volatile("v_mov_b32 %0, %1" : "=v"(lo) : "v"(x));
				long long a = acc[m & 3];
				a = (a & 0xFFFFFFFF00000000ll) | (unsigned)lo;
				asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a) : "v"(x), "v"(wgt) : "vcc");
				acc[m & 3] = a;
			}
			else if constexpr (OP == 1)
				asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(facc[m & 3]) : "v"(fx), "v"(fw));
			else
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(iacc[m & 7]) : "v"(x));
		}
	}
	if constexpr (K > 0)
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	for (int q = 0; q < 4; ++q)
		sink ^= (unsigned)acc[q] ^ (unsigned)(acc[q] >> 32) ^ (unsigned)facc[q].x ^ (unsigned)facc[q].y;
	for (int q = 0; q < 8; ++q)
		sink ^= iacc[q];
	if (sink == 0xDEADBEEFu)
		out[tid] = sink;
}

template <int K, int M, int OP, int DEP>
static double run(unsigned *d_out)
{
	const unsigned trips = 4000;
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	(void)hipFuncSetAttribute((const void *)k<K, M, OP, DEP>, hipFuncAttributeMaxDynamicSharedMemorySize, WAVES * 8192);
	hipLaunchKernelGGL((k<K, M, OP, DEP>), 256, WAVES * 64, WAVES * 8192, 0, 200u, d_out);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL((k<K, M, OP, DEP>), 256, WAVES * 64, WAVES * 8192, 0, trips, d_out);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, e0, e1);
	return ms * 1e6 / trips;   // ns per trip (all 16 waves of a CU make one trip each in that time)
}

template <int K, int M, int OP>
static void triple(const char *op, unsigned *d_out)
{
	const double both = run<K, M, OP, 0>(d_out), dep = run<K, M, OP, 1>(d_out), reads = run<K, 0, OP, 0>(d_out), math = run<0, M, OP, 0>(d_out);
	printf("%-14s %2d reads + %3d ops per trip: reads alone %7.1f ns, arithmetic alone %7.1f ns, both %7.1f ns (max %.1f, sum %.1f), arithmetic fed by the reads %7.1f ns\n", op, K, M,
	       reads, math, both, reads > math ? reads : math, reads + math, dep);
}

int main()
{
	unsigned *d_out;
	(void)hipMalloc(&d_out, 4096 * 4);
	triple<15, 30, 0>("mov+mad_i64", d_out);
	triple<15, 15, 0>("mov+mad_i64", d_out);
	triple<15, 60, 0>("mov+mad_i64", d_out);
	triple<8, 30, 0>("mov+mad_i64", d_out);
	triple<15, 30, 1>("pk_fma_f32", d_out);
	triple<15, 60, 1>("pk_fma_f32", d_out);
	triple<15, 60, 2>("v_add_u32", d_out);
	triple<15, 120, 2>("v_add_u32", d_out);
	return 0;
}
