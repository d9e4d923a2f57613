// dmarepack.hip - can LDS-DMA repack frames of an ODD channel count while it copies them?  Every lane of `buffer_load_dwordx3 ... lds` /
// `buffer_load_dwordx4 ... lds` has its own source offset, so lane i can fetch HALF a frame - 12 of the 22 bytes of an 11-channel
// frame, 16 of the 30 bytes of a 15-channel one - and the halves land back to back in LDS at 12 / 16 bytes per lane: frames padded
// to an even number of dwords (the second half drags the first samples of the next frame along: phantom channels), every lane's
// share dword-aligned.  Checked here: the bytes that land (sources at 2-byte alignment, the descriptor's base at 4), and that
// reads past the descriptor's range come back as zeros.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int SIZE>
__global__ void k(const unsigned char *in, unsigned first_byte, unsigned frame_bytes, unsigned records, unsigned char *out)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const unsigned aligned = first_byte & ~3u;
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(in) + aligned, 0, (int)records, 0x00020000);
	const unsigned lane = threadIdx.x;
	const unsigned off = (lane / 2u) * frame_bytes + (lane % 2u) * SIZE + (first_byte - aligned);
	if constexpr (SIZE == 12)   // (the builtin wants a literal)
		__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)smem, 12, (int)off, 0, 0, 0);
	else
		__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)smem, 16, (int)off, 0, 0, 0);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	for (unsigned b = lane; b < 64u * SIZE; b += 64u)
		out[b] = smem[b];
}

template <int SIZE>
int run(unsigned frame_bytes, unsigned first_byte, unsigned valid_frames)
{
	std::vector<unsigned char> h(8192);
	for (size_t i = 0; i < h.size(); ++i)
		h[i] = (unsigned char)(i * 37u + 11u);
	unsigned char *d_in, *d_out;
	CHECK(hipMalloc(&d_in, h.size()));
	CHECK(hipMalloc(&d_out, 64 * SIZE));
	CHECK(hipMemcpy(d_in, h.data(), h.size(), hipMemcpyHostToDevice));
	// range: valid_frames frames from first_byte, counted in whole dwords from the 4-byte aligned base
	const unsigned aligned = first_byte & ~3u;
	const unsigned records = ((first_byte - aligned) + valid_frames * frame_bytes + 3u) & ~3u;
	k<SIZE><<<1, 64, 64 * SIZE>>>(d_in, first_byte, frame_bytes, records, d_out);
	CHECK(hipDeviceSynchronize());
	std::vector<unsigned char> got(64 * SIZE);
	CHECK(hipMemcpy(got.data(), d_out, got.size(), hipMemcpyDeviceToHost));
	int bad = 0;
	for (unsigned lane = 0; lane < 64; ++lane)
		for (unsigned b = 0; b < (unsigned)SIZE; ++b)
		{
			const unsigned rel = (first_byte - aligned) + (lane / 2u) * frame_bytes + (lane % 2u) * SIZE + b;   // offset from the descriptor's base
			// the range check works per dword of the access: a dword that starts inside [0, records) is delivered whole
			const unsigned dword_start = rel - ((rel - ((first_byte - aligned) + (lane / 2u) * frame_bytes + (lane % 2u) * SIZE)) % 4u);
			const unsigned char want = dword_start + 4u <= records ? h[aligned + rel] : 0;
			if (got[lane * SIZE + b] != want)
			{
				if (bad < 4)
					printf("   lane %u byte %u: got %u want %u (offset %u of %u)\n", lane, b, got[lane * SIZE + b], want, rel, records);
				++bad;
			}
		}
	printf("%d-byte LDS-DMA, frames of %u bytes from byte %u (2-byte aligned: %s), %u valid frames: %s\n", SIZE, frame_bytes, first_byte, (first_byte & 2u) ? "yes" : "no",
	       valid_frames, bad ? "MISMATCH" : "as expected");
	CHECK(hipFree(d_in));
	CHECK(hipFree(d_out));
	return bad;
}

int main()
{
	int bad = 0;
	bad += run<12>(22, 100, 40);   // 11 channels, dword-aligned start, every frame valid
	bad += run<12>(22, 102, 40);   // ... 2-byte aligned start
	bad += run<12>(22, 102, 20);   // ... the range ends in the middle of the tile: zeros beyond
	bad += run<12>(18, 54, 40);    // 9 channels as 12 slots
	bad += run<16>(30, 100, 40);   // 15 channels
	bad += run<16>(30, 98, 40);
	bad += run<16>(30, 98, 17);
	bad += run<16>(26, 50, 40);    // 13 channels as 16 slots (three phantom channels)
	printf(bad ? "FAILED\n" : "all as expected\n");
	return bad != 0;
}
