// launchbench.hip - what a kernel launch costs on this stack whatever the kernel does: K back-to-back launches of (a) an empty
// kernel, (b) a kernel of 512 workgroups x 1024 threads that only stages 32 KB from L2 into LDS per workgroup (k_poly's rows) and
// ends, timed with one HIP event pair.  The floor under the short-stream numbers of tools/size_sweep.py.
// hipcc --offload-arch=gfx950 -O3 -o launchbench launchbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_empty() {}

__global__ __launch_bounds__(1024) void k_stage(const uint4 *rows, unsigned *sink)
{
	extern __shared__ uint4 lds[];
	for (unsigned i = threadIdx.x; i < 2048u; i += 1024u)
		lds[i] = rows[i];
	__syncthreads();
	if (lds[threadIdx.x].x == 0x12345678u)
		sink[0] = 1;
}

template <typename F>
static double per_launch_us(F launch, int k)
{
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	for (int i = 0; i < 200; ++i)
		launch();
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(e0));
	for (int i = 0; i < k; ++i)
		launch();
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	return ms * 1e3 / k;
}

int main()
{
	uint4 *rows;
	unsigned *sink;
	CHECK(hipMalloc(&rows, 32768));
	CHECK(hipMemset(rows, 0, 32768));
	CHECK(hipMalloc(&sink, 4));
	CHECK(hipFuncSetAttribute((const void *)k_stage, hipFuncAttributeMaxDynamicSharedMemorySize, 66064));
	printf("empty kernel, 1 workgroup of 64 threads          %.2f us per launch\n", per_launch_us([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); }, 2000));
	printf("empty kernel, 512 workgroups of 1024 threads      %.2f us per launch\n", per_launch_us([&] { hipLaunchKernelGGL(k_empty, dim3(512), dim3(1024), 0, 0); }, 2000));
	printf("32 KB of rows into LDS, 1 workgroup (66 KB LDS)    %.2f us per launch\n", per_launch_us([&] { hipLaunchKernelGGL(k_stage, dim3(1), dim3(1024), 66064, 0, rows, sink); }, 2000));
	printf("32 KB of rows into LDS, 512 workgroups             %.2f us per launch\n", per_launch_us([&] { hipLaunchKernelGGL(k_stage, dim3(512), dim3(1024), 66064, 0, rows, sink); }, 2000));
	return 0;
}
