// pinbench.hip - what the host-buffer entry points can hope for: H2D of 106 MB + D2H of 230 MB (cfg 2) from ordinary
// malloc'ed memory, (a) as pageable copies, (b) after hipHostRegister on the caller's buffers (cost of registering and
// unregistering included), (c) pinned chunks overlapped on two streams.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
	const size_t in_bytes = 105840000 + 24, out_bytes = 230400768;
	char *in = (char *)malloc(in_bytes), *out = (char *)malloc(out_bytes);
	memset(in, 1, in_bytes);
	memset(out, 2, out_bytes);
	void *d_in, *d_out;
	CHECK(hipMalloc(&d_in, in_bytes));
	CHECK(hipMalloc(&d_out, out_bytes));
	hipStream_t s0, s1;
	CHECK(hipStreamCreate(&s0));
	CHECK(hipStreamCreate(&s1));
	for (int rep = 0; rep < 3; ++rep)
	{
		double t0 = now();
		CHECK(hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, s0));
		CHECK(hipStreamSynchronize(s0));
		double t1 = now();
		CHECK(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, s0));
		CHECK(hipStreamSynchronize(s0));
		double t2 = now();
		printf("pageable: H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)  total %.2f ms\n", (t1 - t0) * 1e3, in_bytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3, out_bytes / (t2 - t1) / 1e9, (t2 - t0) * 1e3);
	}
	for (int rep = 0; rep < 3; ++rep)
	{
		double t0 = now();
		CHECK(hipHostRegister(in, in_bytes, hipHostRegisterDefault));
		CHECK(hipHostRegister(out, out_bytes, hipHostRegisterDefault));
		double t1 = now();
		CHECK(hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, s0));
		CHECK(hipStreamSynchronize(s0));
		double t2 = now();
		CHECK(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, s0));
		CHECK(hipStreamSynchronize(s0));
		double t3 = now();
		CHECK(hipHostUnregister(in));
		CHECK(hipHostUnregister(out));
		double t4 = now();
		printf("registered: register %.2f ms  H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)  unregister %.2f ms  total %.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3,
		       in_bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3, out_bytes / (t3 - t2) / 1e9, (t4 - t3) * 1e3, (t4 - t0) * 1e3);
	}
	for (int rep = 0; rep < 3; ++rep)
	{
		// both directions at once (PCIe is full duplex): registered buffers, two streams
		CHECK(hipHostRegister(in, in_bytes, hipHostRegisterDefault));
		CHECK(hipHostRegister(out, out_bytes, hipHostRegisterDefault));
		double t0 = now();
		CHECK(hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, s0));
		CHECK(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, s1));
		CHECK(hipStreamSynchronize(s0));
		CHECK(hipStreamSynchronize(s1));
		double t1 = now();
		CHECK(hipHostUnregister(in));
		CHECK(hipHostUnregister(out));
		printf("registered, both directions at once: %.2f ms (%.1f GB/s aggregate)\n", (t1 - t0) * 1e3, (in_bytes + out_bytes) / (t1 - t0) / 1e9);
	}
	for (int rep = 0; rep < 3; ++rep)
	{
		// the same from PAGEABLE memory, one host thread: do the async calls return before the copy is done?
		double t0 = now();
		CHECK(hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, s0));
		double t1 = now();
		CHECK(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, s1));
		double t2 = now();
		CHECK(hipStreamSynchronize(s0));
		CHECK(hipStreamSynchronize(s1));
		double t3 = now();
		printf("pageable, both directions at once: H2D call %.2f ms, D2H call %.2f ms, all done after %.2f ms (%.1f GB/s aggregate)\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t0) * 1e3,
		       (in_bytes + out_bytes) / (t3 - t0) / 1e9);
	}
	for (int rep = 0; rep < 3; ++rep)
	{
		// pipelined chunks, pageable, one host thread, two streams: chunk c = H2D(c) then D2H(c) on stream c % 2 (what
		// cr_run_host would do with two workspaces)
		const int chunks = 8;
		const size_t ci = in_bytes / chunks & ~(size_t)15, co = out_bytes / chunks & ~(size_t)15;
		double t0 = now();
		for (int c = 0; c < chunks; ++c)
		{
			hipStream_t s = (c & 1) ? s1 : s0;
			CHECK(hipMemcpyAsync((char *)d_in + c * ci, in + c * ci, ci, hipMemcpyHostToDevice, s));
			CHECK(hipMemcpyAsync(out + c * co, (char *)d_out + c * co, co, hipMemcpyDeviceToHost, s));
		}
		double t1 = now();
		CHECK(hipStreamSynchronize(s0));
		CHECK(hipStreamSynchronize(s1));
		double t2 = now();
		printf("pageable, 8 chunks on two streams: calls returned after %.2f ms, all done after %.2f ms (%.1f GB/s aggregate)\n", (t1 - t0) * 1e3, (t2 - t0) * 1e3, (ci + co) * chunks / (t2 - t0) / 1e9);
	}
	for (int rep = 0; rep < 3; ++rep)
	{
		// pageable, TWO host threads: one uploads while the other downloads (8 chunks each)
		const int chunks = 8;
		const size_t ci = in_bytes / chunks & ~(size_t)15, co = out_bytes / chunks & ~(size_t)15;
		double t0 = now();
		std::thread down([&] {
			CHECK(hipSetDevice(0));
			for (int c = 0; c < chunks; ++c)
			{
				CHECK(hipMemcpyAsync(out + c * co, (char *)d_out + c * co, co, hipMemcpyDeviceToHost, s1));
				CHECK(hipStreamSynchronize(s1));
			}
		});
		for (int c = 0; c < chunks; ++c)
		{
			CHECK(hipMemcpyAsync((char *)d_in + c * ci, in + c * ci, ci, hipMemcpyHostToDevice, s0));
			CHECK(hipStreamSynchronize(s0));
		}
		double t1 = now();
		down.join();
		double t2 = now();
		printf("pageable, two host threads: uploads done after %.2f ms, everything after %.2f ms (%.1f GB/s aggregate)\n", (t1 - t0) * 1e3, (t2 - t0) * 1e3, (ci + co) * chunks / (t2 - t0) / 1e9);
	}
	// a single host thread copying into / out of a pinned bounce buffer
	{
		char *bounce;
		CHECK(hipHostMalloc((void **)&bounce, 64 << 20, hipHostMallocDefault));
		double t0 = now();
		for (size_t o = 0; o + (64 << 20) <= out_bytes; o += 64 << 20)
			memcpy(out + o, bounce, 64 << 20);
		double t1 = now();
		printf("host memcpy pinned -> pageable, one thread: %.1f GB/s\n", (out_bytes / (64 << 20)) * (double)(64 << 20) / (t1 - t0) / 1e9);
		CHECK(hipHostFree(bounce));
	}
	return 0;
}
