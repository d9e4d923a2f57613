// occbench.hip - how many workgroups of a persistent grid are really resident per CU on MI355X, as a function of workgroup
// size and dynamic LDS: every workgroup stamps its start, idles ~20 us, and the host counts those that started in the first
// 5 us.  (Found while reading k_poly's clock stamps: with 66 KB of LDS two 1024-thread workgroups per CU did not co-reside.)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MODE 1: a workgroup barrier in the loop; MODE 2: ~40 live VGPRs; MODE 3: both; MODE 4: __launch_bounds__(1024) and both
template <int MODE>
__global__ void kx(unsigned long long *stamps, unsigned hold_ticks)
{
	extern __shared__ unsigned char smem[];
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	int r[40];
	if (MODE & 2)
	{
#pragma unroll
		for (int i = 0; i < 40; ++i)
			r[i] = threadIdx.x * (i + 1);
	}
	if (threadIdx.x == 0)
		smem[0] = 1;
	while (__builtin_amdgcn_s_memrealtime() - t0 < hold_ticks)
	{
		__builtin_amdgcn_s_sleep(8);
		if (MODE & 1)
			__syncthreads();
		if (MODE & 2)
		{
#pragma unroll
			for (int i = 0; i < 40; ++i)
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) % 40]));
		}
	}
	int x = 0;
	if (MODE & 2)
	{
#pragma unroll
		for (int i = 0; i < 40; ++i)
			x ^= r[i];
	}
	if (threadIdx.x == 0)
		stamps[blockIdx.x] = t0 + (x == 0x7fffffff ? 1 : 0);
}

__global__ void k(unsigned long long *stamps, unsigned hold_ticks)
{
	extern __shared__ unsigned char smem[];
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	if (threadIdx.x == 0)
		smem[0] = 1;
	while (__builtin_amdgcn_s_memrealtime() - t0 < hold_ticks)
		__builtin_amdgcn_s_sleep(8);
	if (threadIdx.x == 0)
		stamps[blockIdx.x] = t0;
}

int main()
{
	unsigned long long *d;
	const int blocks_max = 4096;
	CHECK(hipMalloc(&d, blocks_max * 8));
	const int threads_list[] = {1024, 768, 512, 256};
	const int lds_list[] = {0, 16 << 10, 32 << 10, 48 << 10, 64 << 10, 65 << 10, 66064, 72 << 10, 80 << 10, 81920 - 256, 96 << 10, 128 << 10, 160 << 10};
	for (int threads : threads_list)
		for (int lds : lds_list)
		{
			if (hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
				continue;
			const int blocks = 256 * 8;
			CHECK(hipMemset(d, 0, blocks_max * 8));
			hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, d, 2000u);   // 20 us at 100 MHz
			if (hipDeviceSynchronize() != hipSuccess) { printf("threads %d lds %d: launch failed\n", threads, lds); (void)hipGetLastError(); continue; }
			std::vector<unsigned long long> h(blocks);
			CHECK(hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost));
			const unsigned long long first = *std::min_element(h.begin(), h.end());
			int early = 0;
			for (auto t : h)
				early += (t - first) < 500;   // within 5 us of the first
			printf("threads %4d  lds %6d B: %4d of %d workgroups started within 5 us = %.2f per CU\n", threads, lds, early, blocks, early / 256.0);
		}
	// the same with a barrier / register pressure, at the geometry of k_poly's headline instance
	auto probe = [&](const char *what, const void *fn, void (*launch)(unsigned long long *, int, int)) {
		for (int lds : {0, 66064})
		{
			if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
				continue;
			for (int blocks : {512, 2048})
			{
				CHECK(hipMemset(d, 0, blocks_max * 8));
				launch(d, blocks, lds);
				CHECK(hipDeviceSynchronize());
				std::vector<unsigned long long> h(blocks);
				CHECK(hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost));
				const unsigned long long first = *std::min_element(h.begin(), h.end());
				int early = 0;
				for (auto t : h)
					early += (t - first) < 500;
				printf("%-28s threads 1024  lds %6d B  grid %4d: %4d started within 5 us = %.2f per CU\n", what, lds, blocks, early, early / 256.0);
			}
		}
	};
	for (int lds : {0, 32768, 65536, 66064, 81920, 98304})
	{
		int n = -1;
		hipFuncAttributes attr;
		CHECK(hipFuncSetAttribute((const void *)kx<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
		CHECK(hipFuncGetAttributes(&attr, (const void *)kx<3>));
		const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)kx<3>, 1024, lds);
		printf("runtime occupancy of kx<3> at 1024 threads, %6d B dynamic LDS: rc %d, %d per CU (numRegs %d, static LDS %zu, max dynamic %d)\n", lds, (int)e, n, attr.numRegs, attr.sharedSizeBytes, attr.maxDynamicSharedSizeBytes);
	}
	probe("barrier", (const void *)kx<1>, [](unsigned long long *p, int b, int l) { hipLaunchKernelGGL(kx<1>, dim3(b), dim3(1024), l, 0, p, 2000u); });
	probe("40 VGPRs", (const void *)kx<2>, [](unsigned long long *p, int b, int l) { hipLaunchKernelGGL(kx<2>, dim3(b), dim3(1024), l, 0, p, 2000u); });
	probe("barrier + 40 VGPRs", (const void *)kx<3>, [](unsigned long long *p, int b, int l) { hipLaunchKernelGGL(kx<3>, dim3(b), dim3(1024), l, 0, p, 2000u); });
	return 0;
}
