// cr_kernels.hip - the windowed-sinc hot path on CDNA4 (gfx950) + the thin C-ABI launch shim (crhip.h).
//
// What runs here is the body of the reference's per-output-frame loop: ClownResampler_LowestLevel_Resample
// (reference clownresampler.h:986-1035) evaluated for every output frame that
// ClownResampler_LowLevel_Resample (clownresampler.h:1058-1092) would walk to.  Output frames are independent:
// frame j sits at the 16.16 position pos0 + j * increment (closed form of clownresampler.h:1076-1078).
//
// Two kernels:
//
//  k_poly     The fast path.  The host (cr_plan.c) re-indexes the caller's Lanczos table into POLYPHASE ROWS:
//             every fractional position maps to one row holding the `slots` weights that position uses
//             (zero-padded to a common window) followed by the exact 17.15 reciprocal of their sum
//             (clownresampler.h:1025), so the device does neither the strided table walk nor the integer divide.
//             A persistent workgroup stages the rows in LDS once, then streams its contiguous block of output
//             frames tile by tile: the input PCM window of a tile is fetched with 16-byte buffer loads into
//             registers while the previous tile is being computed, parked in a double-buffered LDS tile, and each
//             lane produces whole output frames from LDS (weights: ds_read_b128 of its row; samples: one LDS read
//             per tap covering all channels of the frame).  Per tap and channel the arithmetic is
//             v_mul_i32_i24 + truncate-toward-zero /65536 + add, in that order: the reference truncates every
//             product BEFORE accumulating (clownresampler.h:1020), which is what rules out dot-product
//             instructions and MFMA.  All of it is 32-bit: the host only selects this kernel when it has proved
//             the bounds (|weight| < 2^23, |acc| < 2^23, |acc * reciprocal| < 2^31).
//
//  k_generic  The reference arithmetic restated with 64-bit integers, one lane per output frame, weights read
//             from the original table in global memory.  It takes every configuration the reference accepts and
//             the accumulate-into semantics of ClownResampler_LowestLevel_Resample; it is what runs when k_poly's
//             preconditions do not hold, and it doubles as an independent second implementation in the tests.
//
// gfx950 only.  No CUDA/HIP dual paths.

#include <hip/hip_runtime.h>

#include <stdint.h>
#include <string.h>

#include "crhip.h"

namespace
{

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------------
// Fixed-point pieces
// ---------------------------------------------------------------------------------------------------------

// (sample * weight) / 65536 with C semantics (truncation toward zero), clownresampler.h:1020 via :625.
// Both operands fit 24 bits (|sample| <= 2^15; |weight| < 2^23 checked by the host), the product fits int32
// (-32768 * 65536 is exactly INT32_MIN), so the full-rate 24-bit multiplier is exact.
__device__ __forceinline__ int tap_term(int sample, int weight)
{
	const int product = __mul24(sample, weight);
	return (product + (int)((unsigned)(product >> 31) >> 16)) >> 16;
}

// (acc * reciprocal) / 32768 with C semantics, clownresampler.h:1033.  Host-proved: |acc| < 2^23,
// 0 < reciprocal < 2^23 and either |acc * reciprocal| < 2^31 (NORM_S31) or < 2^32 (NORM_U32: the product of the
// magnitudes is exact in the low 32 bits of the 24-bit multiplier; truncation toward zero is symmetric in sign).
template <int NORM>
__device__ __forceinline__ int normalise(int acc, int reciprocal)
{
	if constexpr (NORM == CRHIP_NORM_S31)
	{
		const int product = __mul24(acc, reciprocal);
		return (product + (int)((unsigned)(product >> 31) >> 17)) >> 15;
	}
	else
	{
		const int sign = acc >> 31;
		const unsigned magnitude = (unsigned)((acc ^ sign) - sign);
		const unsigned quotient = __umul24(magnitude, (unsigned)reciprocal) >> 15;
		return ((int)quotient ^ sign) - sign;
	}
}

// One input frame (CH interleaved int16) from LDS -> CH ints.
template <int CH>
struct FrameLoad
{
	static __device__ __forceinline__ void load(const unsigned char *p, int (&s)[CH])
	{
#pragma unroll
		for (int c = 0; c < CH; ++c)
			s[c] = reinterpret_cast<const short *>(p)[c];
	}
};

template <>
struct FrameLoad<2>
{
	static __device__ __forceinline__ void load(const unsigned char *p, int (&s)[2])
	{
		const int d = *reinterpret_cast<const int *>(p);
		s[0] = (int)(short)d;
		s[1] = d >> 16;
	}
};

template <>
struct FrameLoad<4>
{
	static __device__ __forceinline__ void load(const unsigned char *p, int (&s)[4])
	{
		const i32x2 d = *reinterpret_cast<const i32x2 *>(p);
		s[0] = (int)(short)d.x;
		s[1] = d.x >> 16;
		s[2] = (int)(short)d.y;
		s[3] = d.y >> 16;
	}
};

template <>
struct FrameLoad<8>
{
	static __device__ __forceinline__ void load(const unsigned char *p, int (&s)[8])
	{
		const i32x4 d = *reinterpret_cast<const i32x4 *>(p);
		s[0] = (int)(short)d.x;
		s[1] = d.x >> 16;
		s[2] = (int)(short)d.y;
		s[3] = d.y >> 16;
		s[4] = (int)(short)d.z;
		s[5] = d.z >> 16;
		s[6] = (int)(short)d.w;
		s[7] = d.w >> 16;
	}
};

// One output frame (CH int32) -> global memory, widest stores the frame size allows.
template <int CH>
__device__ __forceinline__ void store_frame(int *dst, const int (&v)[CH])
{
	if constexpr (CH % 4 == 0)
	{
#pragma unroll
		for (int c = 0; c < CH; c += 4)
		{
			i32x4 q;
			q.x = v[c];
			q.y = v[c + 1];
			q.z = v[c + 2];
			q.w = v[c + 3];
			*reinterpret_cast<i32x4 *>(dst + c) = q;
		}
	}
	else if constexpr (CH % 2 == 0)
	{
#pragma unroll
		for (int c = 0; c < CH; c += 2)
		{
			i32x2 q;
			q.x = v[c];
			q.y = v[c + 1];
			*reinterpret_cast<i32x2 *>(dst + c) = q;
		}
	}
	else
	{
#pragma unroll
		for (int c = 0; c < CH; ++c)
			dst[c] = v[c];
	}
}

// ---------------------------------------------------------------------------------------------------------
// Row index of a fractional position (host mirror: cr_plan.c cr_plan_row_of)
// ---------------------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ unsigned row_of(const crhip_poly_launch &a, unsigned frac)
{
	if constexpr (MODE == CRHIP_ROWMODE_UPSAMPLE)
	{
		return (65536u - frac) >> 6;
	}
	else
	{
		// min_relative / max_relative of clownresampler.h:993-994, kernel_start of :1001
		const unsigned mr = (frac + a.delta + 65535u) >> 16;
		const unsigned xr = (frac + a.skr) >> 16;
		const unsigned kstart = __umul24(a.step, (mr << 16) - frac) >> 16;
		return (unsigned)((int)kstart + a.aff_a * (int)mr + a.aff_b * (int)xr + a.aff_c);
	}
}

// ---------------------------------------------------------------------------------------------------------
// k_poly
// ---------------------------------------------------------------------------------------------------------
// CH        channels (compile time)
// TT        slots when > 0 (fully unrolled, weights in registers); 0 = run-time slot count
// MODE      row-index formula
// NORM      final normalisation form (CRHIP_NORM_*)
// NTHREADS  workgroup size
// NV        16-byte input vectors each thread moves per tile (LDS tile buffer = NV * 16 * NTHREADS bytes)
template <int CH, int TT, int MODE, int NORM, int NTHREADS, int NV>
__global__ __launch_bounds__(NTHREADS) void k_poly(const crhip_poly_launch a)
{
	constexpr unsigned FB = CH * 2;                       // bytes per input frame
	constexpr unsigned TILE_BYTES = NV * 16u * NTHREADS;
	constexpr int RS_CT = (TT + 1 + 3) & ~3;              // row stride when TT is fixed

	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const unsigned tid = threadIdx.x;
	const unsigned rows_bytes = (a.rows * a.row_stride * 4u + 15u) & ~15u;
	const int *rows = reinterpret_cast<const int *>(smem);
	unsigned char *tiles = smem + rows_bytes;

	// contiguous block of output frames owned by this workgroup
	const uint64_t j_begin = (uint64_t)blockIdx.x * a.frames_per_block;
	if (j_begin >= a.n_out)
		return;
	const uint64_t j_end = (j_begin + a.frames_per_block < a.n_out) ? j_begin + a.frames_per_block : a.n_out;

	// stage the polyphase rows once per workgroup (L2-resident after the first workgroups)
	{
		const unsigned nvec = rows_bytes / 16u;
		const u32x4 *src = reinterpret_cast<const u32x4 *>(a.d_rows);
		u32x4 *dst = reinterpret_cast<u32x4 *>(smem);
		for (unsigned i = tid; i < nvec; i += NTHREADS)
			dst[i] = src[i];
	}

	const uint64_t in_base = reinterpret_cast<uint64_t>(a.d_in);
	const uint64_t in_end = in_base + a.in_valid_bytes;
	const unsigned T = TT > 0 ? (unsigned)TT : a.slots;
	const unsigned RS = TT > 0 ? (unsigned)RS_CT : a.row_stride;

	// Issues the global loads of the input window of the tile starting at output frame jt into registers.
	// Returns the byte offset of the window's first frame inside the (16-byte aligned) tile image.
	u32x4 pre[NV];
	auto fetch = [&](uint64_t jt, unsigned n) -> unsigned {
		const uint64_t pos = a.pos0 + jt * (uint64_t)a.increment;
		const uint64_t first_byte = in_base + ((pos >> 16) + a.first_slot) * FB;
		const uint64_t aligned = first_byte & ~(uint64_t)15;
		const unsigned shift = (unsigned)(first_byte - aligned);
		// bytes of the window: frames [0, last_rel + T) where last_rel is the last frame's integer advance
		const unsigned last_rel = (unsigned)(((pos & 0xFFFFu) + (uint64_t)(n - 1) * a.increment) >> 16);
		uint64_t want = (uint64_t)shift + (uint64_t)(last_rel + T) * FB;
		uint64_t avail = in_end > aligned ? in_end - aligned : 0;
		if (want > avail)
			want = avail;
		// wave-uniform descriptor: base = aligned window start, num_records = bytes we may touch
		const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)aligned);
		const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(aligned >> 32));
		const unsigned rec = __builtin_amdgcn_readfirstlane((unsigned)want);
		const __amdgpu_buffer_rsrc_t rsrc =
		    __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0, (int)rec, 0x00020000);
#pragma unroll
		for (int v = 0; v < NV; ++v)
			pre[v] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((v * NTHREADS + tid) * 16u), 0, 0);
		return shift;
	};
	auto park = [&](unsigned char *tile) {
#pragma unroll
		for (int v = 0; v < NV; ++v)
			*reinterpret_cast<u32x4 *>(tile + (v * NTHREADS + tid) * 16u) = pre[v];
	};

	const unsigned NT = a.tile_frames;
	uint64_t jt = j_begin;
	unsigned n = (unsigned)((j_end - jt < NT) ? (j_end - jt) : NT);
	unsigned shift = fetch(jt, n);
	park(tiles);
	__syncthreads();

	for (unsigned it = 0;; ++it)
	{
		const unsigned char *tile = tiles + (it & 1u) * TILE_BYTES;
		const uint64_t jn = jt + n;
		const bool more = jn < j_end;
		unsigned n_next = 0, shift_next = 0;

		if (more)
		{
			n_next = (unsigned)((j_end - jn < NT) ? (j_end - jn) : NT);
			shift_next = fetch(jn, n_next);   // in flight while this tile is computed
		}

		const uint64_t pos = a.pos0 + jt * (uint64_t)a.increment;
		const unsigned frac0 = (unsigned)(pos & 0xFFFFu);
		int *out_tile = reinterpret_cast<int *>(a.d_out) + jt * CH;

		for (unsigned jl = tid; jl < n; jl += NTHREADS)
		{
			const unsigned rel = __umul24(jl, a.increment) + frac0;   // 16.16 relative to the tile's first integer position
			const unsigned frac = rel & 0xFFFFu;
			const unsigned row = row_of<MODE>(a, frac);
			const unsigned char *src = tile + shift + (rel >> 16) * FB;
			const int *wrow = rows + row * RS;

			int acc[CH];
#pragma unroll
			for (int c = 0; c < CH; ++c)
				acc[c] = 0;

			int reciprocal;

			if constexpr (TT > 0)
			{
				int w[RS_CT];
#pragma unroll
				for (int q = 0; q < RS_CT; q += 4)
				{
					const i32x4 v = *reinterpret_cast<const i32x4 *>(wrow + q);
					w[q] = v.x;
					w[q + 1] = v.y;
					w[q + 2] = v.z;
					w[q + 3] = v.w;
				}
#pragma unroll
				for (int s = 0; s < TT; ++s)
				{
					int smp[CH];
					FrameLoad<CH>::load(src + s * FB, smp);
#pragma unroll
					for (int c = 0; c < CH; ++c)
						acc[c] += tap_term(smp[c], w[s]);
				}
				reciprocal = w[TT];
			}
			else
			{
				for (unsigned s = 0; s < T; ++s)
				{
					const int weight = wrow[s];
					int smp[CH];
					FrameLoad<CH>::load(src + s * FB, smp);
#pragma unroll
					for (int c = 0; c < CH; ++c)
						acc[c] += tap_term(smp[c], weight);
				}
				reciprocal = wrow[T];
			}

			int outv[CH];
#pragma unroll
			for (int c = 0; c < CH; ++c)
				outv[c] = normalise<NORM>(acc[c], reciprocal);
			store_frame<CH>(out_tile + (size_t)jl * CH, outv);
		}

		if (!more)
			break;

		park(tiles + ((it + 1u) & 1u) * TILE_BYTES);
		__syncthreads();
		jt = jn;
		n = n_next;
		shift = shift_next;
	}
}

// ---------------------------------------------------------------------------------------------------------
// k_generic - reference arithmetic, 64-bit, one lane per output frame (clownresampler.h:986-1035)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_generic(const crhip_generic_launch a)
{
	const uint64_t j = (uint64_t)blockIdx.x * 256u + threadIdx.x;
	if (j >= a.n_out)
		return;

	const uint64_t fr = a.pos_frac + j * a.increment;           // clownresampler.h:1076-1078, j times
	const uint64_t pos_int = a.pos_int + (fr >> 16);
	const uint64_t pos_frac = fr & 0xFFFFu;

	const uint64_t first_rel = (pos_frac + a.delta + 65535u) >> 16;       // :993
	const uint64_t last_rel = (pos_frac + a.skr) >> 16;                   // :994
	const uint64_t first_frame = pos_int + first_rel;                     // :995
	const uint64_t end_frame = pos_int + a.radius_frames + last_rel;      // :996
	uint64_t table_at = (a.step * ((first_rel << 16) - pos_frac)) >> 16;  // :1001

	const short *in = reinterpret_cast<const short *>(a.d_in);
	const unsigned ch = a.channels;

	long long acc[CRHIP_MAX_CHANNELS];
#pragma unroll
	for (int c = 0; c < CRHIP_MAX_CHANNELS; ++c)
		acc[c] = (a.d_acc_in != nullptr && c < (int)ch) ? a.d_acc_in[c] : 0;

	long long weight_sum = 0;

	for (uint64_t f = first_frame; f < end_frame; ++f, table_at += a.step)
	{
		const long long weight = table_at < a.table_len ? a.d_table[table_at] : 0;   // :1012 asserts the index in range
		const short *src = in + f * ch;
		weight_sum += weight;                                                          // :1016
#pragma unroll
		for (int c = 0; c < CRHIP_MAX_CHANNELS; ++c)
			if (c < (int)ch)
				acc[c] += (long long)src[c] * weight / 65536;                          // :1020
	}

	// :1025 - the host refuses configurations whose weight sum can be 0 (the reference traps there)
	const long long reciprocal = weight_sum != 0 ? 2147483648ll / weight_sum : 0;

#pragma unroll
	for (int c = 0; c < CRHIP_MAX_CHANNELS; ++c)
	{
		if (c < (int)ch)
		{
			const long long v = acc[c] * reciprocal / 32768;                           // :1033
			if (a.out64)
				reinterpret_cast<long long *>(a.d_out)[j * ch + c] = v;
			else
				reinterpret_cast<int *>(a.d_out)[j * ch + c] = (int)v;
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// Instance table of k_poly
// ---------------------------------------------------------------------------------------------------------
constexpr int POLY_THREADS = 256;
constexpr int POLY_VECS = 2;

typedef void (*poly_fn)(const crhip_poly_launch);

template <int CH, int TT>
poly_fn pick_mode(uint32_t mode, uint32_t norm)
{
	if (norm == CRHIP_NORM_S31)
		return mode == CRHIP_ROWMODE_UPSAMPLE ? (poly_fn)k_poly<CH, TT, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, POLY_THREADS, POLY_VECS>
		                                      : (poly_fn)k_poly<CH, TT, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, POLY_THREADS, POLY_VECS>;
	return mode == CRHIP_ROWMODE_UPSAMPLE ? (poly_fn)k_poly<CH, TT, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32, POLY_THREADS, POLY_VECS>
	                                      : (poly_fn)k_poly<CH, TT, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_U32, POLY_THREADS, POLY_VECS>;
}

template <int CH>
poly_fn pick_slots(uint32_t slots, uint32_t mode, uint32_t norm, bool specialised)
{
	if (specialised)
	{
		switch (slots)
		{
			case 5: return pick_mode<CH, 5>(mode, norm);
			case 6: return pick_mode<CH, 6>(mode, norm);
			case 7: return pick_mode<CH, 7>(mode, norm);
			case 15: return pick_mode<CH, 15>(mode, norm);
			case 16: return pick_mode<CH, 16>(mode, norm);
			default: break;
		}
	}
	return pick_mode<CH, 0>(mode, norm);
}

poly_fn pick_poly(uint32_t channels, uint32_t slots, uint32_t mode, uint32_t norm, bool specialised)
{
	switch (channels)
	{
		case 1: return pick_slots<1>(slots, mode, norm, specialised);
		case 2: return pick_slots<2>(slots, mode, norm, specialised);
		case 3: return pick_slots<3>(slots, mode, norm, false);
		case 4: return pick_slots<4>(slots, mode, norm, specialised);
		case 5: return pick_slots<5>(slots, mode, norm, false);
		case 6: return pick_slots<6>(slots, mode, norm, false);
		case 7: return pick_slots<7>(slots, mode, norm, false);
		case 8: return pick_slots<8>(slots, mode, norm, specialised);
		default: return nullptr;
	}
}

bool slots_specialised(uint32_t channels, uint32_t slots)
{
	const bool ch_ok = channels == 1 || channels == 2 || channels == 4 || channels == 8;
	const bool slots_ok = slots == 5 || slots == 6 || slots == 7 || slots == 15 || slots == 16;
	return ch_ok && slots_ok;
}

} // namespace

// -------------------------------------------------------------------------------------------------------------
// C-ABI shim
// -------------------------------------------------------------------------------------------------------------
extern "C"
{

const char *crhip_error_string(int code)
{
	return hipGetErrorString((hipError_t)code);
}

int crhip_device_count(int *count)
{
	*count = 0;
	return (int)hipGetDeviceCount(count);
}

int crhip_set_device(int ordinal)
{
	return (int)hipSetDevice(ordinal);
}

int crhip_get_device_info(int ordinal, crhip_device_info *info)
{
	hipDeviceProp_t prop;
	const hipError_t e = hipGetDeviceProperties(&prop, ordinal);
	if (e != hipSuccess)
		return (int)e;
	memset(info, 0, sizeof(*info));
	info->compute_units = prop.multiProcessorCount;
	info->max_lds_per_block = (int)prop.sharedMemPerBlock;
	info->wavefront = prop.warpSize;
	info->clock_khz = prop.clockRate;
	info->total_memory = prop.totalGlobalMem;
	strncpy(info->name, prop.name, sizeof(info->name) - 1);
	strncpy(info->arch, prop.gcnArchName, sizeof(info->arch) - 1);
	return 0;
}

int crhip_malloc(void **device_pointer, size_t bytes)
{
	return (int)hipMalloc(device_pointer, bytes);
}

int crhip_free(void *device_pointer)
{
	return (int)hipFree(device_pointer);
}

int crhip_host_alloc(void **host_pointer, size_t bytes)
{
	return (int)hipHostMalloc(host_pointer, bytes, hipHostMallocDefault);
}

int crhip_host_free(void *host_pointer)
{
	return (int)hipHostFree(host_pointer);
}

int crhip_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
	return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
}

int crhip_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
	return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream);
}

int crhip_memset(void *dst, int value, size_t bytes, void *stream)
{
	return (int)hipMemsetAsync(dst, value, bytes, (hipStream_t)stream);
}

int crhip_stream_create(void **stream)
{
	hipStream_t s = nullptr;
	const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	*stream = (void *)s;
	return (int)e;
}

int crhip_stream_destroy(void *stream)
{
	return (int)hipStreamDestroy((hipStream_t)stream);
}

int crhip_stream_sync(void *stream)
{
	return (int)hipStreamSynchronize((hipStream_t)stream);
}

int crhip_poly_has_instance(uint32_t channels, uint32_t slots, uint32_t row_mode)
{
	(void)row_mode;
	return slots_specialised(channels, slots) ? 1 : 0;
}

void crhip_poly_geometry(uint32_t channels, uint32_t slots, uint32_t *threads, uint32_t *vecs)
{
	(void)channels;
	(void)slots;
	*threads = POLY_THREADS;
	*vecs = POLY_VECS;
}

int crhip_launch_poly(const crhip_poly_launch *launch, void *stream)
{
	const bool specialised = launch->specialised && slots_specialised(launch->channels, launch->slots);
	const poly_fn fn = pick_poly(launch->channels, launch->slots, launch->row_mode, launch->norm_mode, specialised);

	if (fn == nullptr || launch->threads != POLY_THREADS || launch->vecs != POLY_VECS)
		return (int)hipErrorInvalidValue;
	if (launch->n_out == 0)
		return 0;

	if (launch->lds_bytes > 48u * 1024u)
	{
		const hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)launch->lds_bytes);
		if (e != hipSuccess)
			return (int)e;
	}

	hipLaunchKernelGGL(fn, dim3(launch->blocks), dim3(launch->threads), launch->lds_bytes, (hipStream_t)stream, *launch);
	return (int)hipGetLastError();
}

int crhip_launch_generic(const crhip_generic_launch *launch, void *stream)
{
	if (launch->n_out == 0)
		return 0;
	if (launch->channels == 0 || launch->channels > CRHIP_MAX_CHANNELS)
		return (int)hipErrorInvalidValue;

	const uint64_t blocks = (launch->n_out + 255u) / 256u;
	if (blocks > 0x7FFFFFFFull)
		return (int)hipErrorInvalidValue;

	hipLaunchKernelGGL(k_generic, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *launch);
	return (int)hipGetLastError();
}

} // extern "C"
