// dmaalign.hip - what alignment does LDS-DMA (buffer_load_dwordx4 ... lds) need on gfx950?
//   case A: global source only 4-byte aligned (base + 4 k), LDS destination 16-byte aligned
//   case B: global source 16-byte aligned, LDS destination only 4-byte aligned (M0 base + 4 k)
//   case C: global source only 2-byte aligned (mono PCM)
// Each case copies 1 KiB per wave-instruction and the host compares with memcpy.  Also times A against the aligned form.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__global__ void k_copy(const unsigned char *src, unsigned src_off, unsigned lds_off, unsigned char *out, int pieces)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const unsigned lane = threadIdx.x & 63u;
	const unsigned long long addr = (unsigned long long)(src + src_off);
	const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)addr), hi = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32));
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, pieces * 1024, 0x00020000);
	for (int v = 0; v < pieces; ++v)
		__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(smem + lds_off + v * 1024u), 16, (int)(v * 1024u + lane * 16u), 0, 0, 0);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	for (int i = (int)lane; i < pieces * 1024; i += 64)
		out[i] = smem[lds_off + i];
}

__global__ void k_time(const unsigned char *src, unsigned src_off, unsigned *sink, int reps, size_t span)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	unsigned char *mine = smem + wave * 8192u;
	unsigned acc = 0;
	for (int r = 0; r < reps; ++r)
	{
		const unsigned long long addr = (unsigned long long)(src + src_off) + ((size_t)(blockIdx.x * 16 + wave + (size_t)r * gridDim.x * 16) * 8192u) % span;
		const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)addr), hi = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32));
		const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, 8192, 0x00020000);
#pragma unroll
		for (int v = 0; v < 8; ++v)
			__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(mine + v * 1024u), 16, (int)(v * 1024u + lane * 16u), 0, 0, 0);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		acc += *(volatile unsigned *)(mine + lane * 4u);
	}
	if (acc == 0x12345678u)
		sink[0] = acc;
}

int main()
{
	const size_t N = 1 << 20;
	unsigned char *h = (unsigned char *)malloc(N), *d, *dout, *hout = (unsigned char *)malloc(8192);
	for (size_t i = 0; i < N; ++i)
		h[i] = (unsigned char)((i * 2654435761u) >> 13);
	hipMalloc(&d, N);
	hipMalloc(&dout, 8192);
	hipMemcpy(d, h, N, hipMemcpyHostToDevice);
	for (int cs = 0; cs < 3; ++cs)
		for (unsigned k = 0; k < 8; ++k)
		{
			const unsigned src_off = cs == 0 ? 4 * k : (cs == 2 ? 2 * k + 2 : 64), lds_off = cs == 1 ? 4 * k : 0;
			hipMemset(dout, 0xEE, 8192);
			k_copy<<<1, 64, 8192 + 64>>>(d, src_off, lds_off, dout, 4);
			hipError_t e = hipDeviceSynchronize();
			hipMemcpy(hout, dout, 4096, hipMemcpyDeviceToHost);
			int bad = 0;
			for (int i = 0; i < 4096; ++i)
				bad += hout[i] != h[src_off + i];
			printf("case %c k=%u src_off=%u lds_off=%u: %s (%d bytes differ) %s\n", "ABC"[cs], k, src_off, lds_off, bad ? "WRONG" : "ok", bad, e == hipSuccess ? "" : hipGetErrorString(e));
		}
	// timing: 256 workgroups x 16 waves, 8 KiB per wave per rep, over a 512 MiB span
	const size_t span = (size_t)512 << 20;
	unsigned char *big;
	unsigned *sink;
	hipMalloc(&big, span + 65536);
	hipMalloc(&sink, 4);
	hipMemset(big, 1, span + 65536);
	for (unsigned off = 0; off <= 12; off += 4)
		for (int rep = 0; rep < 2; ++rep)
		{
			hipEvent_t a, b;
			hipEventCreate(&a);
			hipEventCreate(&b);
			hipEventRecord(a);
			k_time<<<512, 1024, 16 * 8192>>>(big, off, sink, 64, span);
			hipEventRecord(b);
			hipEventSynchronize(b);
			float ms;
			hipEventElapsedTime(&ms, a, b);
			printf("src offset %2u: %.1f us, %.2f TB/s\n", off, ms * 1e3, 512.0 * 16 * 64 * 8192 / (ms * 1e-3) / 1e12);
		}
	return 0;
}
