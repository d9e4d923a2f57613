// cr_kernels.hip - the windowed-sinc hot path on CDNA4 (gfx950) + the thin C-ABI launch shim (crhip.h).
//
// What runs here is the body of the reference's per-output-frame loop: ClownResampler_LowestLevel_Resample
// (reference clownresampler.h:986-1035) evaluated for every output frame that
// ClownResampler_LowLevel_Resample (clownresampler.h:1058-1092) would walk to.  Output frames are independent:
// frame j sits at the 16.16 position pos0 + j * increment (closed form of clownresampler.h:1076-1078).
//
// Three kernels:
//
//  k_poly     The fast path.  The host (cr_plan.c) re-indexes the caller's Lanczos table into POLYPHASE ROWS:
//             every fractional position maps to one row holding the `slots` weights that position uses
//             (zero-padded to a common window) followed by the exact 17.15 reciprocal of their sum
//             (clownresampler.h:1025), so the device does neither the strided table walk nor the integer divide.
//             A persistent workgroup stages the rows in LDS once, then takes tiles of output frames in stream order
//             (the first by its own number, the rest as atomic tickets - or plain round-robin where that measured
//             better): the input PCM window of the NEXT tile is fetched by LDS-DMA (16-byte `buffer_load ... lds`,
//             bounds-checked by a per-tile buffer descriptor) into the other half of a double-buffered LDS tile while
//             the current tile is computed; each lane produces whole output frames from LDS (weights: ds_read_b128
//             of its row; samples: one LDS read per tap covering all channels of the frame; the reads of frame i+1
//             are issued before the arithmetic of frame i) and stores them non-temporally.  Per tap and channel the
//             arithmetic is v_mul_i32_i24 + truncate-toward-zero /65536 + add, in that order: the reference
//             truncates every product BEFORE accumulating (clownresampler.h:1020), which is what rules out
//             dot-product instructions and MFMA.  All of it is 32-bit: the host only selects this kernel when it has
//             proved the bounds (-65536 < weight <= 65536, |acc| < 2^23, |acc * reciprocal| < 2^31 or 2^32).
//
//  k_wave     The same rows and arithmetic without any workgroup barrier after the staging: every wave streams
//             wave-tiles of 256 frames through its own double-buffered 1 KiB of LDS and draws its own tickets.
//             Faster than k_poly where the arithmetic dominates (8-lobe stereo), slower where memory does.
//
//  k_generic  The reference arithmetic restated with 64-bit integers, one lane per output frame, weights read
//             from the original table in global memory.  It takes every configuration the reference accepts and
//             the accumulate-into semantics of ClownResampler_LowestLevel_Resample; it is what runs when k_poly's
//             preconditions do not hold, and it doubles as an independent second implementation in the tests.
//
// gfx950 only.  No CUDA/HIP dual paths.

#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "cr_instances.hpp"

namespace
{

// k_generic - reference arithmetic, 64-bit, one lane per output frame (clownresampler.h:986-1035)
// ---------------------------------------------------------------------------------------------------------
// One output frame, the reference's way (clownresampler.h:986-1035), into d_out[j * channels ...].
__device__ __forceinline__ void generic_frame(const void *d_in, void *d_out, const int32_t *d_table, const int64_t *d_acc_in, uint32_t table_len, unsigned ch, uint32_t out64,
                                              uint64_t j, uint64_t pos_int, uint64_t pos_frac, uint64_t skr, uint64_t radius_frames, uint64_t delta, uint64_t step)
{
	const uint64_t first_rel = (pos_frac + delta + 65535u) >> 16;         // :993
	const uint64_t last_rel = (pos_frac + skr) >> 16;                     // :994
	const uint64_t first_frame = pos_int + first_rel;                     // :995
	const uint64_t end_frame = pos_int + radius_frames + last_rel;        // :996
	uint64_t table_at = (step * ((first_rel << 16) - pos_frac)) >> 16;    // :1001

	const short *in = reinterpret_cast<const short *>(d_in);

	long long acc[CRHIP_MAX_CHANNELS];
#pragma unroll
	for (int c = 0; c < CRHIP_MAX_CHANNELS; ++c)
		acc[c] = (d_acc_in != nullptr && c < (int)ch) ? d_acc_in[c] : 0;

	long long weight_sum = 0;

	for (uint64_t f = first_frame; f < end_frame; ++f, table_at += step)
	{
		const long long weight = table_at < table_len ? d_table[table_at] : 0;   // :1012 asserts the index in range
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
			if (out64 == 2)
				reinterpret_cast<short *>(d_out)[j * ch + c] = (short)(v > 0x7FFF ? 0x7FFF : (v < -0x7FFF ? -0x7FFF : v));
			else if (out64)
				reinterpret_cast<long long *>(d_out)[j * ch + c] = v;
			else
				reinterpret_cast<int *>(d_out)[j * ch + c] = (int)v;
		}
	}
}

__global__ __launch_bounds__(256) void k_generic(const crhip_generic_launch a)
{
	const uint64_t j = (uint64_t)blockIdx.x * 256u + threadIdx.x;
	if (j >= a.n_out)
		return;

	const uint64_t fr = a.pos_frac + j * a.increment;           // clownresampler.h:1076-1078, j times
	generic_frame(a.d_in, a.d_out, a.d_table, a.d_acc_in, a.table_len, a.channels, a.out64, j, a.pos_int + (fr >> 16), fr & 0xFFFFu,
	              a.skr, a.radius_frames, a.delta, a.step);
}

// k_generic_segments - variable rate in ONE launch: output frame j belongs to the segment whose [first_out, next first_out) holds
// it (bisection; the table is a few hundred KB at most and every lane of a wave walks nearly the same path through it), and is
// frame j - first_out of that segment's own timeline walk (clownresampler.h:1052-1056 applied between segments on the host,
// :1076-1078 inside one).  One launch per segment costs ~5 us each however short the segment is (rows staged per launch): 6,000
// segments of a tenth of a second took 31 ms that way.
__global__ __launch_bounds__(256) void k_generic_segments(const crhip_segments_launch a)
{
	const uint64_t j = (uint64_t)blockIdx.x * 256u + threadIdx.x;
	if (j >= a.n_out)
		return;

	uint32_t lo = 0, hi = a.n_segments;     // the segment is in [lo, hi)
	while (hi - lo > 1u)
	{
		const uint32_t mid = (lo + hi) >> 1;
		if (a.d_segments[mid].first_out <= j)
			lo = mid;
		else
			hi = mid;
	}
	const crhip_segment s = a.d_segments[lo];
	const uint64_t fr = s.pos_frac + (j - s.first_out) * s.increment;
	generic_frame(a.d_in, a.d_out, a.d_table, nullptr, a.table_len, a.channels, a.out_s16 ? 2u : 0u, j, s.pos_int + (fr >> 16), fr & 0xFFFFu,
	              s.skr, s.radius_frames, s.delta, s.step);
}

// ---------------------------------------------------------------------------------------------------------
// The instance table, merged from the instance units (cr_inst_*.hip)
// ---------------------------------------------------------------------------------------------------------
const special *specials(int *count)
{
	// sized from the providers (each says how many instances it has when asked with no table): no fixed capacity to outgrow
	static std::vector<special> table;
	static std::once_flag once;
	std::call_once(once, [] {
		int (*const providers[])(void *, int) = {crk::specials_headline, crk::specials_long, crk::specials_long_b, crk::specials_multi_a, crk::specials_multi_b, crk::specials_multi_c, crk::specials_down};
		size_t total = 0;
		for (auto provider : providers)
			total += (size_t)provider(nullptr, 0);
		table.resize(total);
		size_t at = 0;
		for (auto provider : providers)
			at += (size_t)provider(table.data() + at, (int)(total - at));   // (cannot come up short: it was asked a line ago)
	});
	*count = (int)table.size();
	return table.data();
}

poly_fn ablation_instance(int abl)
{
	return (poly_fn)crk::ablation_instance(abl);
}

const special *find_special(uint32_t channels, uint32_t slots, uint32_t mode, uint32_t norm)
{
	int n;
	const special *t = specials(&n);
	for (int i = 0; i < n; ++i)
		if (t[i].channels == channels && t[i].slots == slots && t[i].mode == mode && t[i].norm == norm)
			return &t[i];
	return nullptr;
}


template <int OUT16>
poly_fn pick_runtime_channels(uint32_t channels, uint32_t mode, uint32_t norm, uint32_t padded)
{
	if (channels >= 1 && channels <= 4)
		return (poly_fn)crk::runtime_instance_1_4(channels, mode, norm, OUT16);
	if (channels <= 8)
		return (poly_fn)crk::runtime_instance_5_8(channels, mode, norm, OUT16);
	return (poly_fn)crk::runtime_instance_9_16(channels, mode, norm, OUT16, padded);
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

// DIAGNOSTIC (CLOWNRESAMPLER_AMD_GUARD_MALLOC=1 / 2 in the environment at first use): every device allocation of the library - rows images,
// tables, ticket blocks, staging sets, the segment table - is placed with the virtual-memory API so that it ENDS on the last mapped byte
// (1; 16-byte aligned, so up to 15 bytes of slack) or STARTS on the first (2) of its own mapping, the pages on either side reserved and
// unmapped, and is unmapped when it is freed: a kernel that strays outside an internal buffer, or uses one that has been released, is
// a GPU memory fault at once instead of a read of whatever hipMalloc put next door.  There is no GPU AddressSanitizer on this pool;
// the GPU tests run the suite once this way (tests/test_gpu_guarded.py).
namespace
{
struct guard_rec
{
	void *base;
	size_t reserved, mapped;
	hipMemGenericAllocationHandle_t handle;
	void *first;
};
static std::map<void *, guard_rec> g_guarded;
static std::mutex g_guard_lock;

static int guard_mode()
{
	static const int mode = [] {
		const char *e = getenv("CLOWNRESAMPLER_AMD_GUARD_MALLOC");
		return (e != nullptr && *e != '\0' && *e != '0') ? (*e == '2' ? 2 : 1) : 0;
	}();
	return mode;
}

static int guard_malloc(void **device_pointer, size_t bytes, int mode)
{
	int device = 0;
	hipError_t e = hipGetDevice(&device);
	if (e != hipSuccess)
		return (int)e;
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	prop.location.id = device;
	size_t g = 0;
	e = hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityMinimum);
	if (e != hipSuccess)
		return (int)e;
	if (g < 4096)
		g = 4096;
	guard_rec r = {};
	r.mapped = ((bytes != 0 ? bytes : 1) + g - 1) / g * g;
	r.reserved = r.mapped + 2 * g;
	e = hipMemAddressReserve(&r.base, r.reserved, g, nullptr, 0);
	if (e != hipSuccess)
		return (int)e;
	e = hipMemCreate(&r.handle, r.mapped, &prop, 0);
	if (e != hipSuccess)
	{
		(void)hipMemAddressFree(r.base, r.reserved);
		return (int)e;
	}
	r.first = static_cast<char *>(r.base) + g;
	hipMemAccessDesc desc = {};
	desc.location = prop.location;
	desc.flags = hipMemAccessFlagsProtReadWrite;
	e = hipMemMap(r.first, r.mapped, 0, r.handle, 0);
	if (e == hipSuccess)
		e = hipMemSetAccess(r.first, r.mapped, &desc, 1);
	if (e != hipSuccess)
	{
		(void)hipMemUnmap(r.first, r.mapped);
		(void)hipMemRelease(r.handle);
		(void)hipMemAddressFree(r.base, r.reserved);
		return (int)e;
	}
	char *p = static_cast<char *>(r.first);
	if (mode == 1)
		p += (r.mapped - bytes) & ~(size_t)15;
	*device_pointer = p;
	std::lock_guard<std::mutex> hold(g_guard_lock);
	g_guarded[p] = r;
	return 0;
}
} // namespace

int crhip_malloc(void **device_pointer, size_t bytes)
{
	if (guard_mode() != 0)
		return guard_malloc(device_pointer, bytes, guard_mode());
	return (int)hipMalloc(device_pointer, bytes);
}

int crhip_free(void *device_pointer)
{
	if (guard_mode() != 0 && device_pointer != nullptr)
	{
		guard_rec r;
		{
			std::lock_guard<std::mutex> hold(g_guard_lock);
			const auto it = g_guarded.find(device_pointer);
			if (it == g_guarded.end())
				return (int)hipFree(device_pointer);
			r = it->second;
			g_guarded.erase(it);
		}
		hipError_t e = hipDeviceSynchronize();   // (what hipFree does before it lets memory go)
		const hipError_t u = hipMemUnmap(r.first, r.mapped);
		const hipError_t h = hipMemRelease(r.handle);
		// The address range is NOT given back (hipMemAddressFree): it stays reserved and unmapped for the life of the process, so that a
		// stale pointer faults for good - and because a range that was freed and handed out again showed kernels and the copy engines
		// different contents on this runtime (tools/experiments/r06/vmm_probe.py): a diagnostic mode must not add a hazard of its own.
		return (int)(e != hipSuccess ? e : u != hipSuccess ? u : h);
	}
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

// 0 and *device_alias when [host, host + bytes) lies in ONE block of page-locked host memory that the current device can address
// (hipHostMalloc / hipHostRegister: what a client that "did the right thing" hands the host-pointer entry points); 1 for
// ordinary pageable memory (not an error: the question was legitimate, the runtime's sticky error is cleared).
int crhip_host_alias(const void *host, size_t bytes, void **device_alias)
{
	hipPointerAttribute_t at;
	void *first = nullptr, *last = nullptr;

	*device_alias = nullptr;
	if (hipPointerGetAttributes(&at, host) != hipSuccess || at.type != hipMemoryTypeHost)
	{
		(void)hipGetLastError();
		return 1;
	}
	if (hipHostGetDevicePointer(&first, const_cast<void *>(host), 0) != hipSuccess || first == nullptr
	 || (bytes > 1 && (hipHostGetDevicePointer(&last, const_cast<char *>(static_cast<const char *>(host)) + bytes - 1, 0) != hipSuccess
	                   || static_cast<char *>(last) - static_cast<char *>(first) != static_cast<ptrdiff_t>(bytes - 1))))
	{
		(void)hipGetLastError();
		return 1;
	}
	*device_alias = first;
	return 0;
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

int crhip_device_sync(void)
{
	return (int)hipDeviceSynchronize();
}

int crhip_event_create(void **event)
{
	hipEvent_t e = nullptr;
	const hipError_t r = hipEventCreateWithFlags(&e, hipEventDisableTiming);
	*event = (void *)e;
	return (int)r;
}

int crhip_event_destroy(void *event)
{
	return (int)hipEventDestroy((hipEvent_t)event);
}

int crhip_event_record(void *event, void *stream)
{
	return (int)hipEventRecord((hipEvent_t)event, (hipStream_t)stream);
}

int crhip_event_sync(void *event)
{
	return (int)hipEventSynchronize((hipEvent_t)event);
}

int crhip_stream_wait_event(void *stream, void *event)
{
	return (int)hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0);
}

int crhip_stream_is_capturing(void *stream, int *capturing)
{
	hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
	const hipError_t e = hipStreamIsCapturing((hipStream_t)stream, &status);
	*capturing = (e == hipSuccess && status == hipStreamCaptureStatusActive) ? 1 : 0;
	return (int)e;
}

int crhip_stream_busy(void *stream)
{
	const hipError_t e = hipStreamQuery((hipStream_t)stream);
	if (e == hipErrorNotReady)
		return 1;
	if (e != hipSuccess)
		(void)hipGetLastError();   // a handle the caller has destroyed: not an error of ours, and certainly not busy
	return 0;
}

int crhip_enable_peer_access(int device, int peer)
{
	int can = 0, current = 0;
	hipError_t e = hipGetDevice(&current);
	if (e != hipSuccess)
		return (int)e;
	if (device == peer || hipDeviceCanAccessPeer(&can, device, peer) != hipSuccess || !can)
		return 0;
	e = hipSetDevice(device);
	if (e == hipSuccess)
	{
		e = hipDeviceEnablePeerAccess(peer, 0);
		if (e == hipErrorPeerAccessAlreadyEnabled)
		{
			(void)hipGetLastError();
			e = hipSuccess;
		}
	}
	(void)hipSetDevice(current);
	return (int)e;
}

int crhip_memcpy_peer(void *dst, int dst_device, const void *src, int src_device, size_t bytes, void *stream)
{
	if (dst_device == src_device)
		return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
	return (int)hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, (hipStream_t)stream);
}

int crhip_get_device(int *ordinal)
{
	return (int)hipGetDevice(ordinal);
}

static poly_fn select_poly(const crhip_poly_launch *launch, uint32_t *geo);

uint32_t crhip_poly_runtime_padded_frame_bytes(uint32_t channels)
{
	switch (channels)
	{
		case 9: return runtime_padded_bytes<9>();
		case 10: return runtime_padded_bytes<10>();
		case 11: return runtime_padded_bytes<11>();
		case 12: return runtime_padded_bytes<12>();
		case 13: return runtime_padded_bytes<13>();
		case 14: return runtime_padded_bytes<14>();
		case 15: return runtime_padded_bytes<15>();
		case 16: return runtime_padded_bytes<16>();
		default: return 0u;
	}
}

int crhip_poly_has_dual(const crhip_poly_launch *launch)
{
	uint32_t geo;
	return launch->dual && launch->channels == 2u && launch->specialised && select_poly(launch, &geo) != nullptr && (geo < 100u || geo == 150u) ? 1 : 0;
}

int crhip_poly_has_instance(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode)
{
	return find_special(channels, slots, row_mode, norm_mode) != nullptr ? 1 : 0;
}

// variant actually used by a launch: an explicit one (< VARIANTS), an ablation (>= 1000: headline geometry), or the
// instance's measured default (CRHIP_VARIANT_DEFAULT)
static uint32_t resolve_variant(const special *sp, uint32_t variant, uint32_t out_s16 = 0)
{
	if (sp == nullptr)
		return 0u;
	if (variant == WAVE2_VARIANT || (variant == CRHIP_VARIANT_DEFAULT && sp->default_variant == WAVE2_VARIANT))
	{
		if (sp->wave2 != nullptr)
			return WAVE2_VARIANT;
		variant = sp->default_variant != WAVE2_VARIANT ? sp->default_variant : sp->wave2_fallback;
	}
	if (sp->lite && sp->mad[0] != nullptr)
	{
		// a lite instance with a chain: the default and the chain's ids are the chain, every k_poly variant id its SDWA form
		if (variant == CRHIP_VARIANT_DEFAULT || variant == MAD_VARIANT || variant == MAD_VARIANT + 1u)
			return (!out_s16 || sp->mad16 != nullptr) ? (variant == MAD_VARIANT + 1u ? MAD_VARIANT + 1u : MAD_VARIANT) : sp->lite_variant;
		return sp->lite_variant;
	}
	if (sp->lite)
		return sp->default_variant != WAVE2_VARIANT ? sp->default_variant : sp->wave2_fallback;
	if (variant >= 1008u && variant <= 1013u)
		return sp->up[0] != nullptr ? UP_VARIANT : (sp->wave[0] != nullptr ? WAVE_VARIANT : 13u);   // diagnostic k_up instance
	if (variant == 1007u && sp->wave[0] != nullptr)
		return WAVE_VARIANT;                                  // diagnostic k_wave instance: k_wave geometry
	if (variant >= 1000u && variant < 1008u)
		return 3u;                                            // diagnostic k_poly instances: headline geometry
	if (variant >= MAD_VARIANT + 2u)
		variant = sp->default_variant;
	if (variant >= MAD_VARIANT)
	{
		if (sp->mad[0] != nullptr && (!out_s16 || sp->mad16 != nullptr))
			return variant;
		variant = 13u;
	}
	if (variant >= UP_VARIANT)
	{
		if (sp->up[0] != nullptr)
			return variant;
		variant = sp->wave[0] != nullptr ? WAVE_VARIANT : 13u;
	}
	if (variant >= SPLIT_VARIANT && sp->split[0] == nullptr)
		variant = 13u;
	if (variant >= WAVE_VARIANT && variant < SPLIT_VARIANT && sp->wave[0] == nullptr)
		variant = 13u;
	if (out_s16 && variant >= SPLIT_VARIANT)
		variant = sp->default_variant < WAVE_VARIANT ? sp->default_variant : 13u;
	if (out_s16 && variant < WAVE_VARIANT)
		return sp->default_variant < WAVE_VARIANT ? sp->default_variant : 13u;   // the k_poly int16 form exists for one variant
	return variant;
}

int crhip_poly_swizzled(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode, uint32_t variant)
{
	// instances that stage their rows rotated for the plan's increment (crhip_poly_launch.swizzle): k_wave2, and k_poly with a
	// run-time slot count
	const special *sp = find_special(channels, slots, row_mode, norm_mode);
	if (sp == nullptr)
		return (variant == RT_WAVE2_VARIANT || variant == RT_WAVE2S_VARIANT) ? 1 : RUNTIME_SWZ;
	const uint32_t v = resolve_variant(sp, variant);
	if (v == MAD_VARIANT || v == MAD_VARIANT + 1u)
		return sp->mad_rotated[0] != nullptr ? 2 : 0;   // 2: has a rotated form beside the plain one - a rotation is the plan's choice
	if (sp->lite && v < WAVE_VARIANT && sp->fn_rotated != nullptr)
		return 2;
	return v == WAVE2_VARIANT ? 1 : 0;
}

int crhip_poly_runtime_wave2s(uint32_t channels)
{
	return crk::runtime_wave2s_instance(channels, 0) != nullptr ? 1 : 0;
}

int crhip_poly_runtime_wave2(uint32_t channels, uint32_t row_mode)
{
	return crk::runtime_wave2_instance(channels, row_mode, 0) != nullptr ? 1 : 0;
}

int crhip_poly_variants(void)
{
	return VARIANTS + 13;   // + the two k_wave variants, the four two-lanes-per-frame variants, the two k_up variants, the two 64-bit-chain variants, k_wave2 (30), its run-time-slot form (31) and k_wave2s (32)
}

int crhip_poly_up_negmask(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode, uint32_t *negmask)
{
	const special *sp = find_special(channels, slots, row_mode, norm_mode);
	if (sp == nullptr || (sp->up[0] == nullptr && sp->mad[0] == nullptr))
		return 0;
	*negmask = sp->up_negmask;
	return 1;
}

int crhip_poly_has_up(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode)
{
	const special *sp = find_special(channels, slots, row_mode, norm_mode);
	return sp != nullptr && sp->up[1] != nullptr ? 1 : 0;
}

int crhip_poly_default_is_up(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode)
{
	const special *sp = find_special(channels, slots, row_mode, norm_mode);
	return sp != nullptr && sp->default_variant >= UP_VARIANT && sp->default_variant < MAD_VARIANT ? 1 : 0;
}

uint32_t crhip_poly_mad_safemask(uint32_t slots)
{
	// the slots the 64-bit chain of k_poly (variants 28 / 29) takes at any magnitude up to 65536 (mad_safemask, cr_device.hpp)
	return slots == 5u ? mad_safemask<5>() : (slots == 15u ? mad_safemask<15>() : 0u);
}

int crhip_poly_mad_any_sign(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode)
{
	const special *sp = find_special(channels, slots, row_mode, norm_mode);
	return sp != nullptr && sp->mad[0] != nullptr && sp->mad_any_sign ? 1 : 0;
}

int crhip_poly_default_is_mad(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode)
{
	const special *sp = find_special(channels, slots, row_mode, norm_mode);
	return sp != nullptr && sp->default_variant >= MAD_VARIANT && sp->default_variant < MAD_VARIANT + 2u && sp->mad[0] != nullptr ? 1 : 0;
}

uint32_t crhip_poly_up_fallback_variant(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode)
{
	// below 2x upsampling k_up does not apply: k_wave2 where the instance has one (8-lobe stereo 44.1 -> 48 kHz: 106 us against
	// k_wave's 127), else k_wave, else k_poly
	const special *sp = find_special(channels, slots, row_mode, norm_mode);
	if (sp != nullptr && sp->wave2 != nullptr)
		return WAVE2_VARIANT;
	return sp != nullptr && sp->wave[0] != nullptr ? WAVE_VARIANT : 13u;
}

int crhip_poly_dynamic_default(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode)
{
	const special *sp = find_special(channels, slots, row_mode, norm_mode);
	return sp != nullptr && sp->dynamic_tiles ? 1 : 0;
}

int crhip_poly_wave2_negmask(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode, uint32_t *negmask)
{
	const special *sp = find_special(channels, slots, row_mode, norm_mode);
	if (sp == nullptr || sp->wave2 == nullptr)
		return -1;                      // no k_wave2 form
	*negmask = sp->up_negmask;
	return sp->wave2_fixed_signs ? 1 : 0;   // 1: the plan's slot signs must match *negmask
}

uint32_t crhip_poly_wave2_safemask(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode)
{
	const special *sp = find_special(channels, slots, row_mode, norm_mode);
	return sp != nullptr && sp->wave2 != nullptr ? sp->wave2_safemask : 0u;
}

uint32_t crhip_poly_wave2_fallback_variant(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode)
{
	const special *sp = find_special(channels, slots, row_mode, norm_mode);
	if (sp == nullptr)
		return 13u;
	if (sp->default_variant != WAVE2_VARIANT)
		return sp->default_variant >= UP_VARIANT && sp->default_variant < MAD_VARIANT ? (sp->wave[0] != nullptr ? WAVE_VARIANT : 13u) : sp->default_variant;
	return sp->wave2_fallback;
}

uint32_t crhip_poly_fallback_variant(void)
{
	return 13u;   // a k_poly variant every specialised instance has
}

void crhip_poly_geometry(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode, uint32_t variant,
                         uint32_t *threads, uint32_t *vecs, uint32_t *frames_multiple)
{
	const special *sp = find_special(channels, slots, row_mode, norm_mode);
	const uint32_t v = sp != nullptr ? resolve_variant(sp, variant) : (uint32_t)runtime_geo((int)channels);

	if (sp != nullptr && v == WAVE2_VARIANT)
	{
		// k_wave2: vecs = 150 + (1 KiB DMA pieces per wave-tile); frames_multiple = frames per ticket
		*threads = sp->wave2_waves * 64u;
		*vecs = 150u + sp->wave2_nvw;
		*frames_multiple = 64u * sp->wave2_iter * 4u;
		return;
	}

	if (sp != nullptr && v >= MAD_VARIANT)
	{
		const int g = sp->mad_any_sign ? (int)sp->mad_geo : (sp->lite ? (int)(sp->lite_variant % 5u) : 3);   // (a lite instance's chain has the geometry of its SDWA form)
		*threads = (uint32_t)GEOMETRY[g].threads;
		*vecs = (uint32_t)GEOMETRY[g].vecs;
		*frames_multiple = *threads * sp->mad_frames;
		return;
	}

	if (sp != nullptr && v >= UP_VARIANT)
	{
		// k_up: vecs = 200; frames_multiple = the cap on output frames per wave-tile (the host sizes the wave-tile so that
		// it spans at most 64 input positions)
		*threads = UP_WAVES * 64u;
		*vecs = 200u;
		*frames_multiple = UP_MAX_WAVE_TILE;
		return;
	}

	if (sp != nullptr && v >= SPLIT_VARIANT)
	{
		// two lanes per frame: a group of threads covers half as many frames
		*threads = (uint32_t)GEOMETRY[SPLIT_GEO[v - SPLIT_VARIANT]].threads;
		*vecs = (uint32_t)GEOMETRY[SPLIT_GEO[v - SPLIT_VARIANT]].vecs;
		*frames_multiple = *threads / 2u;
		return;
	}

	if (sp != nullptr && v >= WAVE_VARIANT)
	{
		// k_wave: vecs = 100 + (1 KiB DMA pieces per wave-tile); frames_multiple = frames per ticket
		*threads = WAVE_WAVES * 64u;
		*vecs = 100u + WAVE_NVW;
		*frames_multiple = 64u * WAVE_ITER * 4u;
		return;
	}

	*threads = (uint32_t)GEOMETRY[v % 5].threads;
	*vecs = (uint32_t)GEOMETRY[v % 5].vecs;
	*frames_multiple = *threads * (1u << ((v / 5) % 2));
	if (sp == nullptr)
		*frames_multiple /= (uint32_t)runtime_split((int)channels);   // a group of threads covers half as many frames
	else if (sp->lite)
		*frames_multiple /= sp->lite_lanes;
}

// geo: index into GEOMETRY, 100 for k_wave, 200 for k_up
static poly_fn select_poly(const crhip_poly_launch *launch, uint32_t *geo)
{
	const special *sp = launch->specialised ? find_special(launch->channels, launch->slots, launch->row_mode, launch->norm_mode) : nullptr;
	const uint32_t v = resolve_variant(sp, launch->variant, launch->out_s16);
	poly_fn fn;

	if (launch->dual)
	{
		// a MONO stream on this stereo instance (crhip_poly_launch.dual): only the chain form of k_poly has it; nullptr otherwise
		if (sp == nullptr || launch->out_s16 || launch->channels != 2u)
			return nullptr;
		if (v == WAVE2_VARIANT)
		{
			*geo = 150u;
			return sp->wave2_dual;
		}
		if (v != MAD_VARIANT)
			return nullptr;
		*geo = sp->mad_any_sign ? sp->mad_geo : (sp->lite ? sp->lite_variant % 5u : 3u);
		return launch->swizzle != 0 ? sp->mad_dual_rotated : sp->mad_dual;
	}

	if (sp != nullptr && v == WAVE2_VARIANT)
	{
		*geo = 150u;
		if (launch->debug_form >= 1u && launch->debug_form <= 3u && !launch->out_s16 && sp->wave2_forms[launch->debug_form - 1u] != nullptr)
			return sp->wave2_forms[launch->debug_form - 1u];   // (diagnostic build only)
		return launch->out_s16 ? sp->wave2_16 : sp->wave2;
	}

	if (sp == nullptr && launch->variant == RT_WAVE2S_VARIANT && launch->vecs >= 150u)
	{
		*geo = 160u;
		return (poly_fn)crk::runtime_wave2s_instance(launch->channels, launch->out_s16 ? 1 : 0);
	}

	if (sp == nullptr && launch->variant == RT_WAVE2_VARIANT && launch->vecs >= 150u)
	{
		*geo = 150u;
		return (poly_fn)crk::runtime_wave2_instance(launch->channels, launch->row_mode, launch->out_s16 ? 1 : 0);
	}

	if (sp != nullptr && v >= MAD_VARIANT)
	{
		*geo = sp->mad_any_sign ? sp->mad_geo : (sp->lite ? sp->lite_variant % 5u : 3u);
		if (launch->swizzle != 0 && sp->mad_rotated[0] != nullptr)
		{
			if (launch->debug_form >= 1u && launch->debug_form <= 6u && !launch->out_s16 && v == MAD_VARIANT && sp->mad_forms[launch->debug_form - 1u] != nullptr)
				return sp->mad_forms[launch->debug_form - 1u];   // (diagnostic build only)
			return launch->out_s16 ? sp->mad16_rotated : sp->mad_rotated[v - MAD_VARIANT];
		}
		return launch->out_s16 ? sp->mad16 : sp->mad[v - MAD_VARIANT];
	}

	if (sp != nullptr && v >= UP_VARIANT)
	{
		*geo = 200u;
		if (launch->variant >= 1008u && launch->variant <= 1013u && !launch->out_s16 && launch->channels == 2 && launch->slots == 15)
			return ablation_instance((int)(launch->variant - 1000u));
		return launch->out_s16 ? sp->up16 : sp->up[v - UP_VARIANT];
	}

	if (sp != nullptr && v >= SPLIT_VARIANT)
	{
		*geo = (uint32_t)SPLIT_GEO[v - SPLIT_VARIANT];
		return sp->split[v - SPLIT_VARIANT];
	}

	if (sp != nullptr && v >= WAVE_VARIANT)
	{
		fn = launch->out_s16 ? sp->wave16 : sp->wave[v - WAVE_VARIANT];
		if (launch->variant == 1007u)
			fn = ablation_instance(7);
		*geo = 100u;
		return fn;
	}

	if (launch->out_s16)
		fn = sp != nullptr ? sp->fn16 : pick_runtime_channels<1>(launch->channels, launch->row_mode, launch->norm_mode, launch->padded);
	else
		fn = sp != nullptr ? sp->fn[v] : pick_runtime_channels<0>(launch->channels, launch->row_mode, launch->norm_mode, launch->padded);
	if (sp != nullptr && sp->lite && sp->fn_rotated != nullptr && launch->swizzle != 0)
		fn = launch->out_s16 ? sp->fn16_rotated : sp->fn_rotated;

	// debug: variant 1000 + k selects timing-only ablation k of the headline instance (results are wrong by design)
	if (sp != nullptr && launch->variant >= 1000u && launch->variant < 1008u && launch->channels == 2 && launch->slots == 5)
		fn = ablation_instance((int)(launch->variant - 1000u));

	*geo = sp != nullptr ? v % 5 : (uint32_t)runtime_geo((int)launch->channels);
	return fn;
}

// One-time per-instance setup (not legal inside a stream capture): allow more than 48 KiB of dynamic LDS.
int crhip_poly_prepare(const crhip_poly_launch *launch)
{
	uint32_t geo;
	const poly_fn fn = select_poly(launch, &geo);

	if (fn == nullptr)
		return (int)hipErrorInvalidValue;

	// hipFuncSetAttribute costs ~0.2 ms: remember, per device, the largest size each instance has been opened up to, so that
	// a client walking through many ratios (a new plan each) pays it once per instance
	{
		static std::mutex lock;
		static std::map<std::pair<int, const void *>, uint32_t> opened;
		int device = 0;
		const hipError_t e = hipGetDevice(&device);
		if (e != hipSuccess)
			return (int)e;

		std::lock_guard<std::mutex> guard(lock);
		uint32_t &bytes = opened[std::make_pair(device, (const void *)fn)];
		if (bytes >= launch->lds_bytes)
			return 0;
		const hipError_t r = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)launch->lds_bytes);
		if (r == hipSuccess)
			bytes = launch->lds_bytes;
		return (int)r;
	}
}

// What the runtime says about the instance a launch would use: workgroups of launch->threads threads and launch->lds_bytes of
// dynamic LDS that fit one CU at a time, and the instance's register / static LDS footprint.
int crhip_poly_occupancy(const crhip_poly_launch *launch, int *workgroups_per_cu, int *vgprs, int *static_lds)
{
	uint32_t geo;
	const poly_fn fn = select_poly(launch, &geo);

	if (fn == nullptr)
		return (int)hipErrorInvalidValue;

	// The two runtime queries cost ~50 us each, and a variable-rate client makes a plan per distinct ratio (same instance, same shape):
	// remembered per (device, function, threads, LDS) - a sibling plan took 110 us to make, tools/plan_create_rate.py.
	struct answer
	{
		int per_cu, vgprs, static_lds;
	};
	static std::mutex lock;
	static std::map<std::tuple<int, const void *, uint32_t, uint32_t>, answer> known;
	int device = 0;
	hipError_t e = hipGetDevice(&device);
	if (e != hipSuccess)
		return (int)e;
	const auto key = std::make_tuple(device, (const void *)fn, launch->threads, launch->lds_bytes);
	{
		std::lock_guard<std::mutex> guard(lock);
		const auto hit = known.find(key);
		if (hit != known.end())
		{
			*workgroups_per_cu = hit->second.per_cu;
			*vgprs = hit->second.vgprs;
			*static_lds = hit->second.static_lds;
			return 0;
		}
	}

	hipFuncAttributes attr;
	e = hipFuncGetAttributes(&attr, (const void *)fn);
	if (e != hipSuccess)
		return (int)e;
	*vgprs = attr.numRegs;
	*static_lds = (int)attr.sharedSizeBytes;
	e = hipOccupancyMaxActiveBlocksPerMultiprocessor(workgroups_per_cu, (const void *)fn, (int)launch->threads, launch->lds_bytes);
	if (e == hipSuccess)
	{
		std::lock_guard<std::mutex> guard(lock);
		known[key] = answer{*workgroups_per_cu, *vgprs, *static_lds};
	}
	return (int)e;
}

int crhip_launch_poly(const crhip_poly_launch *launch, void *stream)
{
	uint32_t geo;
	const poly_fn fn = select_poly(launch, &geo);

	if (fn == nullptr)
		return (int)hipErrorInvalidValue;
	if (geo == 160u)
	{
		const uint32_t per_instruction = 64u / ((launch->channels + 1u) / 2u);
		if (launch->threads % 64u != 0 || launch->threads > 1024u || launch->vecs < 150u || launch->vecs >= 200u || launch->wave_tile == 0
		    || launch->wave_tile % per_instruction != 0 || launch->tile_frames % launch->wave_tile != 0)
			return (int)hipErrorInvalidValue;
	}
	else if (geo == 150u)
	{
		if (launch->threads % 64u != 0 || launch->vecs < 150u || launch->vecs >= 200u || launch->tile_frames < 64u
		    || (launch->tile_frames & (launch->tile_frames - 1u)) != 0
		    || (launch->row_mode == CRHIP_ROWMODE_UPSAMPLE && launch->plane_rows != UP_PLANE_ROWS))   // (k_wave2 reads the planes of pure-upsampling rows at immediate offsets)
			return (int)hipErrorInvalidValue;
	}
	else if (geo == 200u ? (launch->threads != UP_WAVES * 64u || launch->vecs != 200u || launch->tile_frames % 4u != 0 || launch->tile_frames / 4u > UP_MAX_WAVE_TILE || launch->plane_rows != UP_PLANE_ROWS)
	  : geo == 100u ? (launch->threads != WAVE_WAVES * 64u || launch->vecs != 100u + WAVE_NVW)
	                : (launch->threads != (uint32_t)GEOMETRY[geo].threads || launch->vecs != (uint32_t)GEOMETRY[geo].vecs))
		return (int)hipErrorInvalidValue;
	if (launch->n_out == 0)
		return 0;

	hipLaunchKernelGGL(fn, dim3(launch->blocks), dim3(launch->threads), launch->lds_bytes, (hipStream_t)stream, *launch);
	return (int)hipGetLastError();
}

int crhip_launch_segments(const crhip_segments_launch *launch, void *stream)
{
	if (launch->n_out == 0)
		return 0;
	if (launch->channels == 0 || launch->channels > CRHIP_MAX_CHANNELS || launch->n_segments == 0)
		return (int)hipErrorInvalidValue;

	const uint64_t blocks = (launch->n_out + 255u) / 256u;
	if (blocks > 0x7FFFFFFFull)
		return (int)hipErrorInvalidValue;

	hipLaunchKernelGGL(k_generic_segments, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *launch);
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
