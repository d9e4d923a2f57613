// satomic.hip - do scalar memory atomics (s_atomic_add, returning through lgkmcnt into an SGPR) work on gfx950, and what does one
// round trip cost against a vector atomic (returning through vmcnt into a VGPR)?  hipcc --offload-arch=gfx950 -O3 -o satomic satomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void k_scalar(unsigned *counter, unsigned *out, int draws, unsigned long long *cycles)
{
	const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64u;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < draws; ++i)
	{
		unsigned v = 1;
		asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(counter) : "memory");
		if ((threadIdx.x & 63u) == 0)
			out[wave * draws + i] = v;
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	if ((threadIdx.x & 63u) == 0)
		cycles[wave] = t1 - t0;
}

__global__ void k_vector(unsigned *counter, unsigned *out, int draws, unsigned long long *cycles)
{
	const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64u;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < draws; ++i)
	{
		if ((threadIdx.x & 63u) == 0)
			out[wave * draws + i] = atomicAdd(counter, 1u);
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	if ((threadIdx.x & 63u) == 0)
		cycles[wave] = t1 - t0;
}

static int check(const char *name, std::vector<unsigned> v, const std::vector<unsigned long long> &cyc, int draws)
{
	std::sort(v.begin(), v.end());
	size_t bad = 0;
	for (size_t i = 0; i < v.size(); ++i)
		bad += v[i] != i;
	unsigned long long sum = 0;
	for (auto c : cyc)
		sum += c;
	printf("%-8s %zu draws, %zu not unique/dense, %.0f cycles (100 MHz ticks x?) per draw per wave\n", name, v.size(), bad, (double)sum / cyc.size() / draws);
	return bad != 0;
}

int main()
{
	const int blocks = 256, threads = 256, draws = 64, waves = blocks * threads / 64;
	unsigned *counter, *out;
	unsigned long long *cycles;
	hipMalloc(&counter, 4);
	hipMalloc(&out, waves * draws * 4);
	hipMalloc(&cycles, waves * 8);
	std::vector<unsigned> h(waves * draws);
	std::vector<unsigned long long> c(waves);
	int rc = 0;
	for (int which = 0; which < 2; ++which)
	{
		hipMemset(counter, 0, 4);
		if (which == 0)
			hipLaunchKernelGGL(k_vector, dim3(blocks), dim3(threads), 0, 0, counter, out, draws, cycles);
		else
			hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(threads), 0, 0, counter, out, draws, cycles);
		if (hipDeviceSynchronize() != hipSuccess)
		{
			printf("%s: kernel failed: %s\n", which ? "scalar" : "vector", hipGetErrorString(hipGetLastError()));
			return 2;
		}
		hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
		hipMemcpy(c.data(), cycles, c.size() * 8, hipMemcpyDeviceToHost);
		rc |= check(which ? "scalar" : "vector", h, c, draws);
	}
	return rc;
}
