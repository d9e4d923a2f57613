// scatterwrite.hip - what k_seg's store pattern costs against a contiguous one, nothing else running.
// Both write the same 461 MB (cfg 3's output: 57.6 M frames of 8 bytes) with 3072 waves (256 workgroups of 768), non-temporal 8-byte
// stores, 64 lanes x 8 B per instruction:
//   contiguous  (k_up2-like): a wave-tile is 720 consecutive frames; instruction i of a tile writes frames [64 i, 64 i + 64)
//   segments K  (k_seg-like): 64 segments 65,536 frames apart per block; a tile is K frames of each; one instruction writes 16
//               consecutive frames (a 128-byte line) of four segments, 16 instructions a chunk of 16 frames of all 64
// build: hipcc --offload-arch=gfx950 -O3 -o scatterwrite scatterwrite.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef int i32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(768) void contiguous(i32x2 *out, unsigned long long n_frames, unsigned tile)
{
	const unsigned lane = threadIdx.x & 63u;
	const unsigned long long wave = (unsigned long long)(threadIdx.x >> 6) * gridDim.x + blockIdx.x, waves = (unsigned long long)gridDim.x * 12u;
	const unsigned long long tiles = (n_frames + tile - 1) / tile;
	i32x2 v;
	v.x = (int)lane;
	v.y = (int)wave;
	for (unsigned long long t = wave; t < tiles; t += waves)
	{
		const unsigned long long first = t * tile;
		for (unsigned i = lane; i < tile && first + i < n_frames; i += 64u)
			__builtin_nontemporal_store(v, out + first + i);
	}
}

__global__ __launch_bounds__(768) void segments(i32x2 *out, unsigned long long n_frames, unsigned K, unsigned long long S)
{
	const unsigned lane = threadIdx.x & 63u;
	const unsigned long long wave = (unsigned long long)(threadIdx.x >> 6) * gridDim.x + blockIdx.x, waves = (unsigned long long)gridDim.x * 12u;
	const unsigned long long blocks = (n_frames + 64 * S - 1) / (64 * S), per_seg = S / K, tiles = blocks * per_seg;
	i32x2 v;
	v.x = (int)lane;
	v.y = (int)wave;
	for (unsigned long long t = wave; t < tiles; t += waves)
	{
		const unsigned long long first = (t / per_seg) * 64 * S + (t % per_seg) * K;
		for (unsigned c = 0; c < K; c += 16u)
			for (unsigned i = 0; i < 16u; ++i)
			{
				const unsigned long long f = first + (unsigned long long)(4u * i + (lane >> 4)) * S + c + (lane & 15u);
				if (f < n_frames)
					__builtin_nontemporal_store(v, out + f);
			}
	}
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
// a lane writes ITS OWN segment's line itself: eight 16-byte stores per 16 frames, every instruction 64 different lines
__global__ __launch_bounds__(768) void own_lines(i32x2 *out, unsigned long long n_frames, unsigned K, unsigned long long S)
{
	const unsigned lane = threadIdx.x & 63u;
	const unsigned long long wave = (unsigned long long)(threadIdx.x >> 6) * gridDim.x + blockIdx.x, waves = (unsigned long long)gridDim.x * 12u;
	const unsigned long long blocks = (n_frames + 64 * S - 1) / (64 * S), per_seg = S / K, tiles = blocks * per_seg;
	i32x4 v;
	v.x = (int)lane;
	v.y = (int)wave;
	v.z = v.w = 7;
	for (unsigned long long t = wave; t < tiles; t += waves)
	{
		const unsigned long long first = (t / per_seg) * 64 * S + (t % per_seg) * K + (unsigned long long)lane * S;
		for (unsigned c = 0; c < K; c += 2u)
			if (first + c + 1 < n_frames)
				__builtin_nontemporal_store(v, reinterpret_cast<i32x4 *>(out + first + c));
	}
}

int main()
{
	const unsigned long long n = 57600096ull;
	i32x2 *out;
	hipMalloc(&out, (n + 4096) * 8);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	auto time = [&](const char *name, auto launch) {
		std::vector<float> ms;
		for (int r = 0; r < 12; ++r)
		{
			hipEventRecord(e0);
			launch();
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			float t;
			hipEventElapsedTime(&t, e0, e1);
			ms.push_back(t);
		}
		std::sort(ms.begin(), ms.end());
		printf("%-28s median %7.1f us  min %7.1f us   %6.0f GB/s\n", name, ms[6] * 1e3, ms[0] * 1e3, n * 8 / (ms[6] * 1e-3) / 1e9);
	};
	time("contiguous, tile 720", [&] { hipLaunchKernelGGL(contiguous, 256, 768, 0, 0, out, n, 720u); });
	time("contiguous, tile 64", [&] { hipLaunchKernelGGL(contiguous, 256, 768, 0, 0, out, n, 64u); });
	for (unsigned K : {32u, 64u, 128u, 256u, 1024u})
	{
		char name[64];
		snprintf(name, sizeof name, "segments 65536 apart, K %u", K);
		time(name, [&] { hipLaunchKernelGGL(segments, 256, 768, 0, 0, out, n, K, 65536ull); });
	}
	for (unsigned K : {64u, 128u})
	{
		char name[64];
		snprintf(name, sizeof name, "own lines 16 B/lane, K %u", K);
		time(name, [&] { hipLaunchKernelGGL(own_lines, 256, 768, 0, 0, out, n, K, 65536ull); });
	}
	time("own lines, 1024 apart, K 64", [&] { hipLaunchKernelGGL(own_lines, 256, 768, 0, 0, out, n, 64u, 1024ull); });
	time("segments 1024 apart, K 64", [&] { hipLaunchKernelGGL(segments, 256, 768, 0, 0, out, n, 64u, 1024ull); });
	time("segments 4096 apart, K 64", [&] { hipLaunchKernelGGL(segments, 256, 768, 0, 0, out, n, 64u, 4096ull); });
	return 0;
}
