// streambench.hip - what does MI355X HBM deliver for the hot path's traffic mix (read 105.84 MB of int16 once,
// write 230.4 MB of int32 once), with NO arithmetic, under different access shapes?  Gives the practical ceiling that
// roofline.frac of k_poly should be read against (the 8 TB/s spec peak is not reachable by any kernel).
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/microbench/streambench.hip -o /tmp/streambench && /tmp/streambench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// A: flat grid, every thread moves RV read vectors and WV write vectors (16 B each), fully coalesced
template <int RV, int WV>
__global__ __launch_bounds__(256) void k_flat(const u32x4 *in, size_t in_vecs, u32x4 *out, size_t out_vecs)
{
	const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
	const size_t nthreads = (size_t)gridDim.x * 256;
	u32x4 acc = {0, 0, 0, 0};
#pragma unroll
	for (int r = 0; r < RV; ++r)
	{
		const size_t i = t + r * nthreads;
		if (i < in_vecs)
			acc += in[i];
	}
#pragma unroll
	for (int w = 0; w < WV; ++w)
	{
		const size_t i = t + w * nthreads;
		if (i < out_vecs)
			out[i] = acc + (unsigned)w;
	}
	if constexpr (WV == 0)
		asm volatile("" ::"v"(acc.x), "v"(acc.y), "v"(acc.z), "v"(acc.w));   // the loads stay (cdna_hip_programming.md rule 17)
}

// A2: flat grid, every thread writes one 32-byte output record as two 16-byte stores (lane stride 32 B: the shape of an
// 8-channel int32 frame per lane) or, INTERLEAVED = 1, two fully coalesced 16-byte stores; reads 16 bytes
template <int INTERLEAVED, int NT>
__global__ __launch_bounds__(256) void k_rec32(const u32x4 *in, size_t in_vecs, u32x4 *out, size_t out_recs)
{
	const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
	u32x4 acc = {0, 0, 0, 0};
	if (t < in_vecs)
		acc = in[t];
	if (t < out_recs)
	{
		const size_t wave_base = (t & ~(size_t)63) * 2;   // first 16-byte vector of this wave's 2 KiB
		const size_t lane = t & 63;
		u32x4 *p0 = INTERLEAVED ? out + wave_base + lane : out + t * 2;
		u32x4 *p1 = INTERLEAVED ? out + wave_base + 64 + lane : out + t * 2 + 1;
		if (NT)
		{
			__builtin_nontemporal_store(acc, p0);
			__builtin_nontemporal_store(acc + 1u, p1);
		}
		else
		{
			*p0 = acc;
			*p1 = acc + 1u;
		}
	}
}

// B: persistent workgroups, contiguous block of the streams per workgroup, tile loop (reads via LDS-free registers)
template <int THREADS, int STORE_BYTES>
__global__ __launch_bounds__(THREADS) void k_persist(const u32x4 *in, size_t in_vecs, unsigned *out, size_t out_bytes, unsigned tile_out_bytes)
{
	const size_t per_block_out = (out_bytes / gridDim.x + 4095) & ~(size_t)4095;
	const size_t o_begin = (size_t)blockIdx.x * per_block_out;
	const size_t o_end = o_begin + per_block_out < out_bytes ? o_begin + per_block_out : out_bytes;
	const double ratio = (double)in_vecs * 16.0 / (double)out_bytes;
	u32x4 acc = {0, 0, 0, 0};
	for (size_t o = o_begin; o < o_end; o += tile_out_bytes)
	{
		const size_t tile_end = o + tile_out_bytes < o_end ? o + tile_out_bytes : o_end;
		// input of this tile
		const size_t i0 = (size_t)(o * ratio) / 16, i1 = (size_t)(tile_end * ratio) / 16;
		for (size_t i = i0 + threadIdx.x; i < i1 && i < in_vecs; i += THREADS)
			acc += in[i];
		for (size_t b = o + (size_t)threadIdx.x * STORE_BYTES; b < tile_end; b += (size_t)THREADS * STORE_BYTES)
		{
			if constexpr (STORE_BYTES == 16)
				*(u32x4 *)((char *)out + b) = acc;
			else if constexpr (STORE_BYTES == 8)
				*(u32x2 *)((char *)out + b) = u32x2{acc.x, acc.y};
			else
				*(unsigned *)((char *)out + b) = acc.x;
		}
	}
}

template <typename F>
static double time_us(F launch, int reps = 20)
{
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	std::vector<float> t;
	for (int r = 0; r < 5; ++r)
	{
		launch(0);
		CHECK(hipDeviceSynchronize());
		CHECK(hipEventRecord(e0));
		for (int i = 0; i < reps; ++i)
			launch(i);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		t.push_back(ms * 1000.f / reps);
	}
	std::sort(t.begin(), t.end());
	return t[t.size() / 2];
}

int main()
{
	const size_t in_bytes = 26460006ull * 4, out_bytes = 28800096ull * 8;
	const size_t in_vecs = in_bytes / 16, out_vecs = out_bytes / 16;
	// Rotating buffers whose TOTAL exceeds 1 GB for each direction on its own: the 256 MiB Infinity Cache must not be able to
	// serve (or absorb) a launch that follows another - with 3 x 230 MB the write-only case of round 1 "ran" at 9.4 TB/s.
	const int SETS = 12;   // 12 x 105.8 MB = 1.27 GB of input, 12 x 230.4 MB = 2.76 GB of output
	u32x4 *in[SETS];
	u32x4 *out[SETS];
	for (int s = 0; s < SETS; ++s)
	{
		CHECK(hipMalloc(&in[s], in_bytes + 4096));
		CHECK(hipMalloc(&out[s], out_bytes + 4096));
		CHECK(hipMemset(in[s], s + 1, in_bytes));
	}
	double total = (double)in_bytes + (double)out_bytes;
	printf("traffic per launch: read %.2f MB + write %.2f MB; %d rotating buffer sets (%.2f GB read side, %.2f GB write side)\n", in_bytes / 1e6, out_bytes / 1e6, SETS,
	       SETS * in_bytes / 1e9, SETS * out_bytes / 1e9);
	auto report = [&](const char *name, double us) { printf("%-58s %8.1f us  %7.0f GB/s  (%.3f of 8 TB/s)\n", name, us, total / us / 1e3, total / us / 1e3 / 8000); };

	{
		// write:read vectors = 2.177; flat kernel with RV=1, WV=2 covers out with grid sized by out/2, plus remainder ignored
		const size_t threads = (out_vecs + 1) / 2;
		const unsigned grid = (unsigned)((threads + 255) / 256);
		report("flat, 1 read vec + 2 write vecs per thread (16 B)", time_us([&](int i) { hipLaunchKernelGGL((k_flat<1, 2>), dim3(grid), dim3(256), 0, 0, in[i % SETS], in_vecs, out[i % SETS], out_vecs); }));
		const unsigned grid4 = (unsigned)(((out_vecs + 3) / 4 + 255) / 256);
		report("flat, 2 read vecs + 4 write vecs per thread", time_us([&](int i) { hipLaunchKernelGGL((k_flat<2, 4>), dim3(grid4), dim3(256), 0, 0, in[i % SETS], in_vecs, out[i % SETS], out_vecs); }));
		const unsigned grid8 = (unsigned)(((out_vecs + 7) / 8 + 255) / 256);
		report("flat, 4 read vecs + 8 write vecs per thread", time_us([&](int i) { hipLaunchKernelGGL((k_flat<4, 8>), dim3(grid8), dim3(256), 0, 0, in[i % SETS], in_vecs, out[i % SETS], out_vecs); }));
		total = (double)out_bytes;   // bytes this case really moves
		report("flat, write only (2 vecs per thread), 230.4 MB", time_us([&](int i) { hipLaunchKernelGGL((k_flat<0, 2>), dim3(grid), dim3(256), 0, 0, in[i % SETS], in_vecs, out[i % SETS], out_vecs); }));
		const unsigned gridr = (unsigned)((in_vecs + 255) / 256);
		total = (double)in_bytes;
		report("flat, read only (1 vec per thread), 105.8 MB", time_us([&](int i) { hipLaunchKernelGGL((k_flat<1, 0>), dim3(gridr), dim3(256), 0, 0, in[i % SETS], in_vecs, out[i % SETS], (size_t)0); }));
		total = (double)in_bytes + (double)out_bytes;
	}
	{
		// 8-channel shape (cfg 4): 28.8 M frames of 16 B in, 26.46 M frames of 32 B out
		const size_t in8 = 28800000ull, out8 = 26460260ull;
		u32x4 *i8, *o8;
		CHECK(hipMalloc(&i8, in8 * 16 + 4096));
		CHECK(hipMalloc(&o8, out8 * 32 + 4096));
		const double tot8 = in8 * 16.0 + out8 * 32.0;
		const unsigned g8 = (unsigned)((in8 + 255) / 256);
		auto rep8 = [&](const char *name, double us) { printf("%-58s %8.1f us  %7.0f GB/s  (%.3f of 8 TB/s)\n", name, us, tot8 / us / 1e3, tot8 / us / 1e3 / 8000); };
		rep8("8-ch shape: 2 x 16 B stores per lane, lane stride 32 B", time_us([&](int) { hipLaunchKernelGGL((k_rec32<0, 0>), dim3(g8), dim3(256), 0, 0, i8, in8, o8, out8); }));
		rep8("8-ch shape: same, non-temporal", time_us([&](int) { hipLaunchKernelGGL((k_rec32<0, 1>), dim3(g8), dim3(256), 0, 0, i8, in8, o8, out8); }));
		rep8("8-ch shape: 2 fully coalesced 16 B stores per lane", time_us([&](int) { hipLaunchKernelGGL((k_rec32<1, 0>), dim3(g8), dim3(256), 0, 0, i8, in8, o8, out8); }));
		rep8("8-ch shape: same, non-temporal", time_us([&](int) { hipLaunchKernelGGL((k_rec32<1, 1>), dim3(g8), dim3(256), 0, 0, i8, in8, o8, out8); }));
	}
	for (unsigned blocks : {512u, 1024u, 2048u})
		for (unsigned tile : {32768u, 65536u, 131072u})
		{
			char name[128];
			snprintf(name, sizeof name, "persistent %u x 1024 thr, tile %u KB out, 8 B stores", blocks, tile / 1024);
			report(name, time_us([&](int i) { hipLaunchKernelGGL((k_persist<1024, 8>), dim3(blocks), dim3(1024), 0, 0, in[i % SETS], in_vecs, (unsigned *)out[i % SETS], out_bytes, tile); }));
			snprintf(name, sizeof name, "persistent %u x 1024 thr, tile %u KB out, 16 B stores", blocks, tile / 1024);
			report(name, time_us([&](int i) { hipLaunchKernelGGL((k_persist<1024, 16>), dim3(blocks), dim3(1024), 0, 0, in[i % SETS], in_vecs, (unsigned *)out[i % SETS], out_bytes, tile); }));
		}
	for (unsigned blocks : {2048u, 4096u, 8192u})
	{
		char name[128];
		snprintf(name, sizeof name, "persistent %u x 256 thr, tile 16 KB out, 16 B stores", blocks);
		report(name, time_us([&](int i) { hipLaunchKernelGGL((k_persist<256, 16>), dim3(blocks), dim3(256), 0, 0, in[i % SETS], in_vecs, (unsigned *)out[i % SETS], out_bytes, 16384u); }));
		snprintf(name, sizeof name, "persistent %u x 256 thr, tile 16 KB out, 8 B stores", blocks);
		report(name, time_us([&](int i) { hipLaunchKernelGGL((k_persist<256, 8>), dim3(blocks), dim3(256), 0, 0, in[i % SETS], in_vecs, (unsigned *)out[i % SETS], out_bytes, 16384u); }));
	}
	return 0;
}
