// ldswin.hip - what a wave's WINDOW reads cost the LDS when the lanes' window bases are `step` frames apart (a lane per output
// frame: k_wave2), by read width.  A stereo frame of the expanded window is 8 bytes.
//   b64        one ds_read_b64 per slot (k_wave2 today)
//   b128       one ds_read_b128 per TWO slots from the lane's own base: 8-byte aligned only (is that legal, and what does it cost?)
//   b128a      the same with the base rounded down to 16 bytes (what two copies of the window, one a frame out of phase, would allow)
// Every kind checks what it read against the pattern in LDS.  16 waves per workgroup, one workgroup per CU, reads only.
//   hipcc --offload-arch=gfx950 -O2 -o ldswin ldswin.hip && ./ldswin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVES = 16;
constexpr unsigned PER_WAVE = 4096;   // bytes of window per wave
constexpr int READS = 16;             // reads per trip (immediate offsets)

template <int KIND>
__global__ __launch_bounds__(WAVES * 64) void k(unsigned increment, unsigned lane_map, unsigned trips, unsigned *out, unsigned *bad)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	unsigned *mine = reinterpret_cast<unsigned *>(smem + wave * PER_WAVE);
	for (unsigned i = lane; i < PER_WAVE / 4u; i += 64u)
		mine[i] = i;   // dword i holds i
	__syncthreads();
	const unsigned flane = lane_map ? (((lane & 31u) << 1) | (lane >> 5)) : lane;
	const unsigned frame = (flane * increment) >> 16;
	unsigned at = (unsigned)(uintptr_t)mine + frame * 8u;
	if (KIND == 2)
		at &= ~15u;
	unsigned sink = 0, wrong = 0;
	for (unsigned t = 0; t < trips; ++t)
	{
		if constexpr (KIND == 0)
		{
			i32x2 v[READS];
#pragma unroll
			for (int r = 0; r < READS; ++r)
				asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[r]) : "v"(at), "n"(r * 8));
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int r = 0; r < READS; ++r)
			{
				asm volatile("" : "+v"(v[r]));
				if (t == 0)
				{
					const unsigned want = (at - (unsigned)(uintptr_t)mine) / 4u + 2u * r;
					wrong += ((unsigned)v[r].x != want) + ((unsigned)v[r].y != want + 1u);
				}
				sink ^= (unsigned)v[r].x;
			}
		}
		else
		{
			i32x4 v[READS / 2];
#pragma unroll
			for (int r = 0; r < READS / 2; ++r)
				asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[r]) : "v"(at), "n"(r * 16));
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int r = 0; r < READS / 2; ++r)
			{
				asm volatile("" : "+v"(v[r]));
				if (t == 0)
				{
					const unsigned want = (at - (unsigned)(uintptr_t)mine) / 4u + 4u * r;
					wrong += ((unsigned)v[r].x != want) + ((unsigned)v[r].y != want + 1u) + ((unsigned)v[r].z != want + 2u) + ((unsigned)v[r].w != want + 3u);
				}
				sink ^= (unsigned)v[r].w;
			}
		}
	}
	if (sink == 0xDEADBEEFu)
		out[tid] = sink;
	if (wrong)
		atomicAdd(bad, wrong);
}

template <int KIND>
static void run(const char *name, double step, unsigned lane_map, unsigned *d_out, unsigned *d_bad)
{
	const unsigned increment = (unsigned)(step * 65536.0 + 0.5);
	const unsigned trips = 2000;
	if (((63u * (lane_map ? 1u : 1u) * (unsigned long long)increment) >> 16) * 8u + READS * 8u + 16u > PER_WAVE)
	{
		printf("%-6s step %7.4f: window does not fit\n", name, step);
		return;
	}
	hipMemset(d_bad, 0, 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const int blocks = 256;
	hipLaunchKernelGGL(k<KIND>, blocks, WAVES * 64, WAVES * PER_WAVE, 0, increment, lane_map, 200u, d_out, d_bad);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<KIND>, blocks, WAVES * 64, WAVES * PER_WAVE, 0, increment, lane_map, trips, d_out, d_bad);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	unsigned bad = 0;
	hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
	// per CU: WAVES waves x trips x READS slots
	const double slots = (double)WAVES * trips * READS;
	printf("%-6s step %7.4f lane_map %u: %7.3f ns per wave-slot per CU (%5.2f cycles at 2.4 GHz)   %s\n", name, step, lane_map, ms * 1e6 / slots,
	       ms * 1e6 / slots * 2.4, bad ? "WRONG VALUES" : "values ok");
}

int main()
{
	unsigned *d_out, *d_bad;
	hipMalloc(&d_out, 4096 * 4);
	hipMalloc(&d_bad, 4);
	const double steps[] = {0.0, 0.5, 0.91875, 1.0, 1.0884, 1.5, 2.0, 2.177, 3.0, 5.5125, 6.0};
	for (double s : steps)
		for (unsigned m = 0; m < 2; ++m)
		{
			run<0>("b64", s, m, d_out, d_bad);
			run<1>("b128", s, m, d_out, d_bad);
			run<2>("b128a", s, m, d_out, d_bad);
		}
	return 0;
}
