/*
 * crhip_fake.c - TEST INFRASTRUCTURE: a second implementation of the crhip_* seam (clownresampler_amd/csrc/crhip.h) in plain C,
 * so that the ~5,000 lines of HOST logic of the product - batching, the ticket-block rings, the plan cache, helper threads,
 * streaming windows, callback replay, segments validation, the sharded call - run in a container without a GPU, under
 * AddressSanitizer / UndefinedBehaviorSanitizer and ThreadSanitizer.  It is linked ONLY into
 * tests/hostshim/libcr_hostshim_*.so, never into libclownresampler_amd.so; the product has no CPU path.
 *
 *   device memory   = malloc (so the sanitizers see every byte a "kernel" reads or writes)
 *   streams, events = tokens; every operation completes before it returns
 *   a kernel launch = a scalar C model of what the kernel computes FROM ITS LAUNCH ARGUMENTS:
 *       k_generic / segments  the oracle's frame (oracle/cr_oracle.c: oracle_frame) over the launch's table and configuration
 *       k_poly                the rows image of the launch (cr_plan.c layouts), the device's row-index formula (cr_device.hpp row_of),
 *                             C's truncating 16.16 multiply per tap (clownresampler.h:1020 via :625) and normalisation (:1033); reads beyond
 *                             in_valid_bytes return 0 as the kernels' buffer descriptors make them
 *   instances       = none of the specialised ones: every plan takes the run-time-slot k_poly geometry or k_generic - the host logic
 *                     around them is what is under test, not instance selection (the GPU suite covers that)
 * CRA_FAKE_DEVICES=n in the environment: n devices (default 1) - the multi-device entry point with its per-device contexts.
 *
 * -DFAKE_REAL_INSTANCES (libcr_hostshim_inst*.so): the instance QUERIES - which (channels, slots, row mode) have a specialised
 * kernel, every variant's geometry, k_up2 / k_wave2 / k_seg / k_int shapes and slot signs - are not answered here but by the PRODUCT's own
 * host code: the HIP objects of clownresampler_amd/csrc/build are linked in with their symbols weakened (objcopy), so that this file's memory
 * operations and launches win and the real tables answer everything else.  Every plan then takes the geometry it takes on the GPU - dual mono,
 * brief shapes, k_seg's tiling, k_int's staged rows, padded tiles - and every launch is (a) put through the REAL launch function's own argument
 * checks (called with n_out = 0: it validates, then returns before hipLaunchKernel), (b) VALIDATED: every byte range the kernel may touch
 * - input [d_in, + in_valid_bytes), rows image, ticket block, output - is touched at both ends inside this file's mallocs, so AddressSanitizer
 * flags a range that leaves its allocation or an image that has been freed, and (c) COMPUTED by a scalar model from the launch arguments alone.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "crhip.h"
#include "cr_oracle.h"

#define FAKE_ERROR_INVALID 1
#define FAKE_ERROR_LAUNCH 719

static __thread int t_device = 0;
static unsigned long long g_launches[4];   /* poly, generic, segments, other */

/* ---- failure injection: crhip_fake_fail(kind, n) makes the n-th next operation of that kind fail ONCE with a HIP-like
   error code; 0 disarms.  What the host logic does with a device that says no - at an allocation, a launch, a copy, a synchronise - is
   under test: nothing may leak (LeakSanitizer), no state may move, the entry points hand the failure back (clownresampler.h:746-748). */
enum { FAKE_FAIL_MALLOC = 0, FAKE_FAIL_HOST_ALLOC = 1, FAKE_FAIL_LAUNCH = 2, FAKE_FAIL_COPY = 3, FAKE_FAIL_SYNC = 4, FAKE_FAIL_KINDS = 5 };
#define FAKE_ERROR_INJECTED 2   /* hipErrorOutOfMemory's number */
static int g_fail_in[FAKE_FAIL_KINDS];
static unsigned long long g_failed[FAKE_FAIL_KINDS];

void crhip_fake_fail(int kind, int nth)
{
	if (kind >= 0 && kind < FAKE_FAIL_KINDS)
		__atomic_store_n(&g_fail_in[kind], nth, __ATOMIC_RELAXED);
}

unsigned long long crhip_fake_failed(int kind)
{
	return kind >= 0 && kind < FAKE_FAIL_KINDS ? __atomic_load_n(&g_failed[kind], __ATOMIC_RELAXED) : 0;
}

static int injected(int kind)
{
	int n = __atomic_load_n(&g_fail_in[kind], __ATOMIC_RELAXED);

	while (n > 0)
	{
		if (__atomic_compare_exchange_n(&g_fail_in[kind], &n, n - 1, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED))
		{
			if (n == 1)
			{
				__atomic_fetch_add(&g_failed[kind], 1ull, __ATOMIC_RELAXED);
				return 1;
			}
			return 0;
		}
	}
	return 0;
}

static int device_count_now(void)
{
	const char *e = getenv("CRA_FAKE_DEVICES");
	const int n = e != NULL ? atoi(e) : 1;
	return n < 0 ? 0 : (n > 16 ? 16 : n);
}

const char *crhip_error_string(int code)
{
	return code == 0 ? "no error" : code == FAKE_ERROR_INJECTED ? "fake device: injected failure" : code == FAKE_ERROR_LAUNCH ? "fake device: a launch broke one of its invariants (see stderr)" : "fake device: invalid value";
}

int crhip_device_count(int *count)
{
	*count = device_count_now();
	return 0;
}

int crhip_set_device(int ordinal)
{
	if (ordinal < 0 || ordinal >= device_count_now())
		return FAKE_ERROR_INVALID;
	t_device = ordinal;
	return 0;
}

int crhip_get_device(int *ordinal)
{
	*ordinal = t_device;
	return 0;
}

int crhip_get_device_info(int ordinal, crhip_device_info *info)
{
	if (ordinal < 0 || ordinal >= device_count_now())
		return FAKE_ERROR_INVALID;
	memset(info, 0, sizeof(*info));
	info->compute_units = 256;
	info->max_lds_per_block = 160 * 1024;
	info->wavefront = 64;
	info->clock_khz = 2400000;
	info->total_memory = (size_t)8 << 30;
	snprintf(info->name, sizeof(info->name), "fake device %d (tests/hostshim)", ordinal);
	snprintf(info->arch, sizeof(info->arch), "gfx950:fake");
	return 0;
}

int crhip_malloc(void **device_pointer, size_t bytes)
{
	/* a little slack behind, as hipMalloc's granularity gives: NOT in front - an underrun is a bug worth seeing */
	if (injected(FAKE_FAIL_MALLOC))
	{
		*device_pointer = NULL;
		return FAKE_ERROR_INJECTED;
	}
	*device_pointer = malloc(bytes != 0 ? bytes : 1);
	return *device_pointer != NULL ? 0 : 2;
}

int crhip_free(void *device_pointer)
{
	free(device_pointer);
	return 0;
}

int crhip_host_alloc(void **host_pointer, size_t bytes)
{
	if (injected(FAKE_FAIL_HOST_ALLOC))
	{
		*host_pointer = NULL;
		return FAKE_ERROR_INJECTED;
	}
	*host_pointer = malloc(bytes != 0 ? bytes : 1);
	return *host_pointer != NULL ? 0 : 2;
}

int crhip_host_free(void *host_pointer)
{
	free(host_pointer);
	return 0;
}

int crhip_host_alias(const void *host, size_t bytes, void **device_alias)
{
	/* CRA_FAKE_PAGE_LOCKED=1: every host buffer counts as page-locked and device-addressable at its own address - cr_run_host's direct path
	   (one launch on the caller's own buffers, whatever their alignment and however little lies behind their last byte) */
	const char *e = getenv("CRA_FAKE_PAGE_LOCKED");

	(void)bytes;
	if (e != NULL && *e == '1')
	{
		*device_alias = (void *)host;
		return 0;
	}
	*device_alias = NULL;
	return 1;   /* pageable: the staged path */
}

int crhip_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
	(void)stream;
	if (injected(FAKE_FAIL_COPY))
		return FAKE_ERROR_INJECTED;
	memcpy(dst, src, bytes);
	return 0;
}

int crhip_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
	(void)stream;
	if (injected(FAKE_FAIL_COPY))
		return FAKE_ERROR_INJECTED;
	memcpy(dst, src, bytes);
	return 0;
}

int crhip_memset(void *dst, int value, size_t bytes, void *stream)
{
	(void)stream;
	memset(dst, value, bytes);
	return 0;
}

int crhip_stream_create(void **stream)
{
	*stream = malloc(16);
	return *stream != NULL ? 0 : 2;
}

int crhip_stream_destroy(void *stream)
{
	free(stream);
	return 0;
}

int crhip_stream_sync(void *stream)
{
	(void)stream;
	return injected(FAKE_FAIL_SYNC) ? FAKE_ERROR_INJECTED : 0;
}

int crhip_device_sync(void)
{
	return 0;
}

int crhip_event_create(void **event)
{
	*event = malloc(16);
	return *event != NULL ? 0 : 2;
}

int crhip_event_destroy(void *event)
{
	free(event);
	return 0;
}

int crhip_event_record(void *event, void *stream)
{
	(void)event;
	(void)stream;
	return 0;
}

int crhip_event_sync(void *event)
{
	(void)event;
	return 0;
}

int crhip_stream_wait_event(void *stream, void *event)
{
	(void)stream;
	(void)event;
	return 0;
}

int crhip_stream_is_capturing(void *stream, int *capturing)
{
	(void)stream;
	*capturing = 0;
	return 0;
}

int crhip_stream_busy(void *stream)
{
	(void)stream;
	return 0;
}

int crhip_enable_peer_access(int device, int peer)
{
	(void)device;
	(void)peer;
	return 0;
}

int crhip_memcpy_peer(void *dst, int dst_device, const void *src, int src_device, size_t bytes, void *stream)
{
	(void)dst_device;
	(void)src_device;
	(void)stream;
	memcpy(dst, src, bytes);
	return 0;
}

/* ---- what instances there are ---- */
#ifdef FAKE_REAL_INSTANCES
/* the product's own tables answer (see the head of this file); the real launch functions, renamed by objcopy, serve as argument checkers */
int real_crhip_launch_poly(const crhip_poly_launch *launch, void *stream);
int real_crhip_launch_int(const crhip_int_launch *launch, void *stream);
int real_crhip_launch_seg(const crhip_seg_launch *launch, void *stream);

static int workgroups_per_cu(uint32_t threads, uint32_t lds_bytes)
{
	/* what the runtime's occupancy query says for kernels that are not register-bound: 160 KiB of LDS and 2,048 threads per CU */
	const uint32_t by_lds = lds_bytes != 0 ? (160u * 1024u) / lds_bytes : 8u, by_threads = threads != 0 ? 2048u / threads : 8u;
	const uint32_t n = by_lds < by_threads ? by_lds : by_threads;
	return (int)(n > 8u ? 8u : n);
}

int crhip_int_prepare(uint32_t channels, uint32_t ratio, uint32_t period, uint32_t slots, int *per_cu, int *per_cu_s16)
{
	crhip_int_shape shape;

	if (!crhip_int_instance(channels, ratio, period, slots, &shape))
		return FAKE_ERROR_INVALID;
	*per_cu = workgroups_per_cu(shape.threads, shape.lds_bytes[0]);
	*per_cu_s16 = workgroups_per_cu(shape.threads, shape.lds_bytes[1]);
	return 0;
}

int crhip_seg_prepare(uint32_t channels, uint32_t slots, uint32_t increment, int *per_cu)
{
	uint32_t negmask = 0, threads = 0, lds = 0, chunk = 0;

	if (!crhip_seg_instance(channels, slots, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32, increment, &negmask, &threads, &lds, &chunk))
		return FAKE_ERROR_INVALID;
	*per_cu = workgroups_per_cu(threads, lds);
	return 0;
}

int crhip_poly_prepare(const crhip_poly_launch *launch)
{
	crhip_poly_launch probe = *launch;

	probe.n_out = 0;   /* the real function checks its arguments, selects the instance - and returns before launching */
	return real_crhip_launch_poly(&probe, NULL);
}

int crhip_poly_occupancy(const crhip_poly_launch *launch, int *workgroups, int *vgprs, int *static_lds)
{
	const int code = crhip_poly_prepare(launch);

	if (code != 0)
		return code;
	*workgroups = workgroups_per_cu(launch->threads, launch->lds_bytes);
	*vgprs = 64;
	*static_lds = 0;
	return *workgroups >= 1 ? 0 : FAKE_ERROR_INVALID;
}
#else

int crhip_int_instance(uint32_t channels, uint32_t ratio, uint32_t period, uint32_t slots, crhip_int_shape *shape)
{
	(void)channels; (void)ratio; (void)period; (void)slots; (void)shape;
	return 0;
}

int crhip_int_prepare(uint32_t channels, uint32_t ratio, uint32_t period, uint32_t slots, int *per_cu, int *per_cu_s16)
{
	(void)channels; (void)ratio; (void)period; (void)slots; (void)per_cu; (void)per_cu_s16;
	return FAKE_ERROR_INVALID;
}


int crhip_seg_instance(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode, uint32_t increment, uint32_t *negmask, uint32_t *threads, uint32_t *lds_bytes, uint32_t *chunk)
{
	(void)channels; (void)slots; (void)row_mode; (void)norm_mode; (void)increment; (void)negmask; (void)threads; (void)lds_bytes; (void)chunk;
	return 0;
}

int crhip_seg_prepare(uint32_t channels, uint32_t slots, uint32_t increment, int *per_cu)
{
	(void)channels; (void)slots; (void)increment; (void)per_cu;
	return FAKE_ERROR_INVALID;
}


int crhip_poly_prepare(const crhip_poly_launch *launch)
{
	(void)launch;
	return 0;
}

int crhip_poly_occupancy(const crhip_poly_launch *launch, int *workgroups_per_cu, int *vgprs, int *static_lds)
{
	(void)launch;
	*workgroups_per_cu = 2;
	*vgprs = 64;
	*static_lds = 0;
	return 0;
}

uint32_t crhip_poly_runtime_padded_frame_bytes(uint32_t channels) { (void)channels; return 0; }
int crhip_poly_has_dual(const crhip_poly_launch *launch) { (void)launch; return 0; }
int crhip_poly_has_instance(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode) { (void)channels; (void)slots; (void)row_mode; (void)norm_mode; return 0; }
int crhip_poly_swizzled(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode, uint32_t variant) { (void)channels; (void)slots; (void)row_mode; (void)norm_mode; (void)variant; return 0; }
int crhip_poly_dynamic_default(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode) { (void)channels; (void)slots; (void)row_mode; (void)norm_mode; return 1; }
uint32_t crhip_poly_fallback_variant(void) { return 13u; }
int crhip_poly_up_negmask(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode, uint32_t *negmask) { (void)channels; (void)slots; (void)row_mode; (void)norm_mode; (void)negmask; return 0; }
int crhip_poly_default_is_mad(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode) { (void)channels; (void)slots; (void)row_mode; (void)norm_mode; return 0; }
int crhip_poly_mad_any_sign(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode) { (void)channels; (void)slots; (void)row_mode; (void)norm_mode; return 0; }
int crhip_poly_has_up(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode) { (void)channels; (void)slots; (void)row_mode; (void)norm_mode; return 0; }
int crhip_poly_default_is_up(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode) { (void)channels; (void)slots; (void)row_mode; (void)norm_mode; return 0; }
uint32_t crhip_poly_up_fallback_variant(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode) { (void)channels; (void)slots; (void)row_mode; (void)norm_mode; return 13u; }
uint32_t crhip_poly_mad_safemask(uint32_t slots) { (void)slots; return 0; }
int crhip_poly_wave2_negmask(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode, uint32_t *negmask) { (void)channels; (void)slots; (void)row_mode; (void)norm_mode; (void)negmask; return -1; }
uint32_t crhip_poly_wave2_safemask(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode) { (void)channels; (void)slots; (void)row_mode; (void)norm_mode; return 0; }
uint32_t crhip_poly_wave2_fallback_variant(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode) { (void)channels; (void)slots; (void)row_mode; (void)norm_mode; return 13u; }
int crhip_poly_runtime_wave2(uint32_t channels, uint32_t row_mode) { (void)channels; (void)row_mode; return 0; }
int crhip_poly_runtime_wave2s(uint32_t channels) { (void)channels; return 0; }
int crhip_poly_variants(void) { return 33; }

void crhip_poly_geometry(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode, uint32_t variant, uint32_t *threads, uint32_t *vecs, uint32_t *frames_multiple)
{
	(void)channels; (void)slots; (void)row_mode; (void)norm_mode; (void)variant;
	*threads = 256u;
	*vecs = 2u;
	*frames_multiple = 256u;
}
#endif /* FAKE_REAL_INSTANCES */

/* ---- the launches ---- */

static int32_t clamp_s16(int64_t v)
{
	return (int32_t)(v > 0x7FFF ? 0x7FFF : (v < -0x7FFF ? -0x7FFF : v));
}

/* sample `index` (int16 units) of the launch's input, 0 beyond what the caller said is readable */
static int64_t input_sample(const void *d_in, uint64_t valid_bytes, uint64_t index)
{
	if ((index + 1u) * 2u > valid_bytes)
		return 0;
	return ((const int16_t *)d_in)[index];
}

/* both ends of a byte range the kernel may touch: inside one of this file's mallocs, or AddressSanitizer speaks */
static void touch(const void *pointer, uint64_t bytes)
{
	if (pointer != NULL && bytes != 0)
	{
		const volatile unsigned char *q = (const volatile unsigned char *)pointer;
		(void)q[0];
		(void)q[bytes - 1u];
	}
}

/* the ticket block arrives zeroed and is left zeroed (crhip.h) */
static int tickets_zeroed(const uint32_t *d_tickets)
{
	uint32_t w;

	for (w = 0; w < CRHIP_TICKET_WORDS; ++w)
		if (d_tickets[w] != 0)
		{
			fprintf(stderr, "crhip_fake: ticket block not zeroed at word %u (a block handed to two launches in flight?)\n", w);
			return 0;
		}
	return 1;
}

int crhip_launch_poly(const crhip_poly_launch *l, void *stream)
{
	const int32_t *image = l->d_rows;
	const uint32_t weight_planes = (l->slots + 3u) / 4u;
	/* cr_plan.c cr_poly_device_image: SPLIT (run-time-slot instances) = planes of four weights, then a plane whose first entry is the
	   reciprocal; COMPACT (specialised instances) = the host row as it is - weights, then the reciprocal at [slots] - cut into planes */
	const int compact = l->specialised != 0;
	const uint32_t stream_channels = l->dual ? 1u : l->channels;
	uint64_t j;

	(void)stream;
	if (injected(FAKE_FAIL_LAUNCH))
		return FAKE_ERROR_INJECTED;
	__atomic_fetch_add(&g_launches[0], 1ull, __ATOMIC_RELAXED);

	/* what every k_poly-family launch has to satisfy */
	if (l->n_out == 0 || l->blocks == 0 || l->tile_frames == 0 || l->threads == 0 || l->d_tickets == NULL || l->d_rows == NULL || l->d_out == NULL
	 || l->lds_bytes > 160u * 1024u || l->channels == 0 || l->channels > CRHIP_MAX_CHANNELS || l->plane_rows < l->rows || l->plane_rows % 16u != 0
	 || l->row_stride % 4u != 0 || l->increment >= (1u << 24) || (l->dual != 0 && (l->channels != 2u || l->out_s16 != 0 || l->dual_out_frames != l->n_out || l->dual_valid_frames > l->n_out))
	 || (compact ? l->row_stride < l->slots + 1u : l->row_stride != 4u * (weight_planes + 1u))
#ifndef FAKE_REAL_INSTANCES
	 || l->dual != 0 || l->padded != 0 || l->specialised != 0 || l->swizzle != 0
#endif
	   )
	{
		fprintf(stderr, "crhip_fake: k_poly launch breaks an invariant (n_out %llu blocks %u tile %u threads %u lds %u channels %u rows %u/%u stride %u slots %u dual %u)\n",
		        (unsigned long long)l->n_out, l->blocks, l->tile_frames, l->threads, l->lds_bytes, l->channels, l->rows, l->plane_rows, l->row_stride, l->slots, l->dual);
		return FAKE_ERROR_LAUNCH;
	}
#ifdef FAKE_REAL_INSTANCES
	{
		/* the product's own launch function on the same arguments (n_out = 0: it checks, selects the instance, and returns before launching) */
		crhip_poly_launch probe = *l;
		int code;

		probe.n_out = 0;
		code = real_crhip_launch_poly(&probe, NULL);
		if (code != 0)
		{
			fprintf(stderr, "crhip_fake: the real crhip_launch_poly refuses this launch (%d): %u ch %u slots variant %u threads %u vecs %u tile %u plane_rows %u\n", code, l->channels, l->slots,
			        l->variant, l->threads, l->vecs, l->tile_frames, l->plane_rows);
			return FAKE_ERROR_LAUNCH;
		}
	}
#endif
	/* every range the kernel may touch */
	touch(l->d_in, l->in_valid_bytes);
	touch(l->d_rows, (uint64_t)l->plane_rows * l->row_stride * 4u);
	touch(l->d_tickets, (uint64_t)CRHIP_TICKET_WORDS * 4u);
	touch(l->d_out, l->dual ? ((uint64_t)l->dual_out_frames + l->dual_valid_frames) * 4u : l->n_out * l->channels * (l->out_s16 ? 2u : 4u));
	if (l->dual && (uint64_t)l->dual_in_bytes > l->in_valid_bytes)
	{
		fprintf(stderr, "crhip_fake: dual mono: the second window starts %u bytes in, the input has %llu\n", l->dual_in_bytes, (unsigned long long)l->in_valid_bytes);
		return FAKE_ERROR_LAUNCH;
	}
	if (!tickets_zeroed(l->d_tickets))
		return FAKE_ERROR_LAUNCH;

	for (j = 0; j < l->n_out; ++j)
	{
		const uint64_t pos = l->pos0 + j * (uint64_t)l->increment;
		const uint32_t frac = (uint32_t)(pos & 0xFFFFu);
		uint32_t row, shift = 0, s, c, half;
		int64_t reciprocal;
		uint64_t first;

		if (l->row_mode == CRHIP_ROWMODE_UPSAMPLE)
			row = (65536u - frac) >> 6;
		else
		{
			/* cr_device.hpp row_of<CRHIP_ROWMODE_AFFINE> */
			const uint32_t mr = (frac + l->delta + 65535u) >> 16;
			const uint32_t xr = (frac + l->skr) >> 16;
			const uint32_t kstart = (uint32_t)(((uint64_t)l->step * ((mr << 16) - frac)) >> 16);
			shift = mr - l->first_mr;
			row = (uint32_t)((int32_t)kstart + l->aff_a * (int32_t)mr + l->aff_b * (int32_t)xr + l->aff_c);
		}
		if (row >= l->rows || shift > l->window_extra)
		{
			fprintf(stderr, "crhip_fake: frame %llu: row %u of %u / shift %u of %u\n", (unsigned long long)j, row, l->rows, shift, l->window_extra);
			return FAKE_ERROR_LAUNCH;
		}
		reciprocal = compact ? image[((size_t)(l->slots / 4u) * l->plane_rows + row) * 4u + l->slots % 4u] : image[((size_t)weight_planes * l->plane_rows + row) * 4u];
		first = (pos >> 16) + l->first_slot + shift;
		/* dual mono: the pair's second frame is frame j + dual_out_frames of the MONO stream, its window dual_in_bytes further on, same row */
		for (half = 0; half < (l->dual ? 2u : 1u); ++half)
		{
			const uint64_t base_sample = half ? l->dual_in_bytes / 2u : 0u;

			if (half && j >= l->dual_valid_frames)
				break;
			for (c = 0; c < stream_channels; ++c)
			{
				int64_t acc = 0, out;

				for (s = 0; s < l->slots; ++s)
				{
					const int64_t weight = image[((size_t)(s / 4u) * l->plane_rows + row) * 4u + s % 4u];
					const int64_t sample = input_sample(l->d_in, l->in_valid_bytes, base_sample + (first + s) * stream_channels + c);
					acc += sample * weight / 65536;   /* C: toward zero (clownresampler.h:1020 via :625) */
				}
				out = acc * reciprocal / 32768;       /* :1033 */
				if (l->dual)
					((int32_t *)l->d_out)[j + (half ? l->dual_out_frames : 0u)] = (int32_t)out;
				else if (l->out_s16)
					((int16_t *)l->d_out)[j * l->channels + c] = (int16_t)clamp_s16(out);
				else
					((int32_t *)l->d_out)[j * l->channels + c] = (int32_t)out;
			}
		}
	}
	return 0;
}

#ifdef FAKE_REAL_INSTANCES
/* k_int from its launch arguments: the staged rows of the period (|weight| << 15, or |weight| for the slots that may reach 65536), the
   instance's slot signs, the period's window starts (cr_kint.hpp; host: cr_context.c int_launch_row) */
int crhip_launch_int(const crhip_int_launch *l, void *stream)
{
	crhip_int_shape shape;
	uint64_t j;

	(void)stream;
	if (injected(FAKE_FAIL_LAUNCH))
		return FAKE_ERROR_INJECTED;
	__atomic_fetch_add(&g_launches[3], 1ull, __ATOMIC_RELAXED);
	{
		crhip_int_launch probe = *l;
		probe.n_out = 0;
		if (real_crhip_launch_int(&probe, NULL) != 0 || !crhip_int_instance(l->channels, l->ratio, l->period, l->slots, &shape) || l->period != shape.period
		 || l->n_out == 0 || l->period == 0 || l->period > 4u || l->slots * l->period > CRHIP_INT_MAX_SLOTS || l->d_out == NULL)
		{
			fprintf(stderr, "crhip_fake: k_int launch breaks an invariant (%u ch, ratio %u:%u, %u slots, %u blocks)\n", l->channels, l->ratio, l->period, l->slots, l->blocks);
			return FAKE_ERROR_LAUNCH;
		}
	}
	touch(l->d_in, l->in_valid_bytes);
	touch(l->d_out, l->n_out * l->channels * (l->out_s16 ? 2u : 4u));
	if (l->d_tickets != NULL)
	{
		touch(l->d_tickets, (uint64_t)CRHIP_TICKET_WORDS * 4u);
		if (!tickets_zeroed(l->d_tickets) || l->ticket_tiles == 0)
			return FAKE_ERROR_LAUNCH;
	}
	for (j = 0; j < l->n_out; ++j)
	{
		const uint32_t phase = (uint32_t)(j % l->period);
		const uint64_t first = l->first_frame + (j / l->period) * l->ratio + shape.starts[phase];
		uint32_t c, s;

		for (c = 0; c < l->channels; ++c)
		{
			int64_t acc = 0, out;

			for (s = 0; s < l->slots; ++s)
			{
				const uint32_t at = phase * l->slots + s;
				const int64_t magnitude = ((shape.safemask >> at) & 1u) ? (int64_t)l->w[at] : (int64_t)((uint32_t)l->w[at] >> 15);
				const int64_t weight = ((shape.zeromask >> at) & 1u) ? 0 : (((shape.negmask >> at) & 1u) ? -magnitude : magnitude);

				acc += input_sample(l->d_in, l->in_valid_bytes, (first + s) * l->channels + c) * weight / 65536;
			}
			out = acc * (int64_t)l->reciprocal[phase] / 32768;
			if (l->out_s16)
				((int16_t *)l->d_out)[j * l->channels + c] = (int16_t)clamp_s16(out);
			else
				((int32_t *)l->d_out)[j * l->channels + c] = (int32_t)out;
		}
	}
	return 0;
}

/* k_seg from its launch arguments: the float image of the rows (|weight| / 65536, then 2 x the reciprocal), the instance's slot signs, the
   launch's tiling held to the rules the kernel's 32-bit arithmetic needs (cr_kseg.hpp; host: cr_context.c cr_plan_launch) */
int crhip_launch_seg(const crhip_seg_launch *l, void *stream)
{
	uint32_t negmask = 0, threads = 0, lds = 0, chunk = 0;
	const uint32_t *image = (const uint32_t *)l->d_rows;
	uint64_t j;

	(void)stream;
	if (injected(FAKE_FAIL_LAUNCH))
		return FAKE_ERROR_INJECTED;
	__atomic_fetch_add(&g_launches[3], 1ull, __ATOMIC_RELAXED);
	{
		crhip_seg_launch probe = *l;
		const uint64_t blocks64 = l->seg_frames != 0 ? (l->n_out + 64u * l->seg_frames - 1u) / (64u * l->seg_frames) : 0;

		probe.n_out = 0;
		if (real_crhip_launch_seg(&probe, NULL) != 0 || !crhip_seg_instance(2u, l->slots, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32, l->increment, &negmask, &threads, &lds, &chunk)
		 || l->n_out == 0 || l->d_rows == NULL || l->d_tickets == NULL || l->d_out == NULL || l->increment == 0 || l->increment >= 65536u
		 || l->seg_frames == 0 || (l->seg_frames * l->increment) % 65536u != 0                       /* the lanes of a wave share their fraction */
		 || l->seg_in_frames != (l->seg_frames * l->increment) >> 16
		 || l->seg_frames * 512u >= (1ull << 32) || l->seg_in_frames * 256u >= (1ull << 32)            /* crhip.h: the kernel's 32-bit offsets */
		 || l->tile_frames % chunk != 0 || l->tiles_per_seg != (l->seg_frames + l->tile_frames - 1u) / l->tile_frames
		 || l->n_tiles != blocks64 * l->tiles_per_seg || l->n_tiles >= (1ull << 32) || l->in_valid_bytes >= 0xFFFFFFFCull)
		{
			fprintf(stderr, "crhip_fake: k_seg launch breaks an invariant (n_out %llu, S %llu, D %llu, K %u x %u, %llu tiles, increment %u)\n", (unsigned long long)l->n_out,
			        (unsigned long long)l->seg_frames, (unsigned long long)l->seg_in_frames, l->tile_frames, l->tiles_per_seg, (unsigned long long)l->n_tiles, l->increment);
			return FAKE_ERROR_LAUNCH;
		}
	}
	touch(l->d_in, l->in_valid_bytes);
	touch(l->d_rows, 1025u * 64u);
	touch(l->d_tickets, (uint64_t)CRHIP_TICKET_WORDS * 4u);
	touch(l->d_out, l->n_out * 8u);
	if (!tickets_zeroed(l->d_tickets))
		return FAKE_ERROR_LAUNCH;
	for (j = 0; j < l->n_out; ++j)
	{
		const uint64_t pos = l->pos0 + j * (uint64_t)l->increment;
		const uint32_t row = (65536u - (uint32_t)(pos & 0xFFFFu)) >> 6;
		const uint64_t first = (pos >> 16) + l->first_slot;
		const int64_t reciprocal = (int64_t)(image[row * 16u + 15u] / 2u);
		uint32_t c, s;

		for (c = 0; c < 2u; ++c)
		{
			int64_t acc = 0;

			for (s = 0; s < 15u; ++s)
			{
				float f;
				int64_t weight;

				memcpy(&f, &image[row * 16u + s], sizeof(f));
				weight = (int64_t)(f * 65536.0f);
				if ((negmask >> s) & 1u)
					weight = -weight;
				acc += input_sample(l->d_in, l->in_valid_bytes, (first + s) * 2u + c) * weight / 65536;
			}
			((int32_t *)l->d_out)[j * 2u + c] = (int32_t)(acc * reciprocal / 32768);
		}
	}
	return 0;
}
#else
int crhip_launch_int(const crhip_int_launch *launch, void *stream)
{
	(void)launch; (void)stream;
	fprintf(stderr, "crhip_fake: k_int launched though no instance was offered\n");
	return FAKE_ERROR_LAUNCH;
}

int crhip_launch_seg(const crhip_seg_launch *launch, void *stream)
{
	(void)launch; (void)stream;
	fprintf(stderr, "crhip_fake: k_seg launched though no instance was offered\n");
	return FAKE_ERROR_LAUNCH;
}
#endif

static void table_to_i64(const int32_t *table, uint32_t len, int64_t *out)
{
	uint32_t i;

	for (i = 0; i < len; ++i)
		out[i] = table[i];
}

static void emit_frame(void *d_out, uint32_t out_kind, uint64_t frame, uint32_t channels, const int64_t *acc)
{
	uint32_t c;

	for (c = 0; c < channels; ++c)
	{
		if (out_kind == 1)
			((int64_t *)d_out)[frame * channels + c] = acc[c];
		else if (out_kind == 2)
			((int16_t *)d_out)[frame * channels + c] = (int16_t)clamp_s16(acc[c]);
		else
			((int32_t *)d_out)[frame * channels + c] = (int32_t)acc[c];
	}
}

int crhip_launch_generic(const crhip_generic_launch *g, void *stream)
{
	oracle_config cfg;
	int64_t *table;
	uint64_t j;

	(void)stream;
	if (injected(FAKE_FAIL_LAUNCH))
		return FAKE_ERROR_INJECTED;
	__atomic_fetch_add(&g_launches[1], 1ull, __ATOMIC_RELAXED);
	if (g->n_out == 0 || g->d_table == NULL || g->d_out == NULL || g->channels == 0 || g->channels > CRHIP_MAX_CHANNELS || (g->d_acc_in != NULL && g->n_out != 1) || g->out64 > 2)
	{
		fprintf(stderr, "crhip_fake: k_generic launch breaks an invariant\n");
		return FAKE_ERROR_LAUNCH;
	}
	table = (int64_t *)malloc((size_t)g->table_len * sizeof(int64_t));
	if (table == NULL)
		return 2;
	table_to_i64(g->d_table, g->table_len, table);
	cfg.stretched_radius = g->skr;
	cfg.radius_frames = g->radius_frames;
	cfg.radius_delta = g->delta;
	cfg.table_step = g->step;
	for (j = 0; j < g->n_out; ++j)
	{
		const uint64_t pos = (g->pos_int << 16) + g->pos_frac + j * g->increment;
		int64_t acc[CRHIP_MAX_CHANNELS];
		uint32_t c;

		for (c = 0; c < g->channels; ++c)
			acc[c] = g->d_acc_in != NULL ? g->d_acc_in[c] : 0;
		oracle_frame(&cfg, table, g->table_len, acc, g->channels, (const int16_t *)g->d_in, pos >> 16, pos & 0xFFFFu);
		emit_frame(g->d_out, g->out64, j, g->channels, acc);
	}
	free(table);
	return 0;
}

int crhip_launch_segments(const crhip_segments_launch *l, void *stream)
{
	int64_t *table;
	uint64_t j;
	uint32_t k = 0;

	(void)stream;
	if (injected(FAKE_FAIL_LAUNCH))
		return FAKE_ERROR_INJECTED;
	__atomic_fetch_add(&g_launches[2], 1ull, __ATOMIC_RELAXED);
	if (l->n_out == 0 || l->n_segments == 0 || l->d_segments == NULL || l->d_table == NULL || l->d_segments[0].first_out != 0)
	{
		fprintf(stderr, "crhip_fake: segments launch breaks an invariant\n");
		return FAKE_ERROR_LAUNCH;
	}
	table = (int64_t *)malloc((size_t)l->table_len * sizeof(int64_t));
	if (table == NULL)
		return 2;
	table_to_i64(l->d_table, l->table_len, table);
	for (j = 0; j < l->n_out; ++j)
	{
		const crhip_segment *seg;
		oracle_config cfg;
		int64_t acc[CRHIP_MAX_CHANNELS] = {0};
		uint64_t pos;

		while (k + 1 < l->n_segments && l->d_segments[k + 1].first_out <= j)
		{
			if (l->d_segments[k + 1].first_out <= l->d_segments[k].first_out)
			{
				fprintf(stderr, "crhip_fake: segment table not ascending / a segment is empty\n");
				free(table);
				return FAKE_ERROR_LAUNCH;
			}
			++k;
		}
		seg = &l->d_segments[k];
		cfg.stretched_radius = seg->skr;
		cfg.radius_frames = seg->radius_frames;
		cfg.radius_delta = seg->delta;
		cfg.table_step = seg->step;
		pos = (seg->pos_int << 16) + seg->pos_frac + (j - seg->first_out) * seg->increment;
		oracle_frame(&cfg, table, l->table_len, acc, l->channels, (const int16_t *)l->d_in, pos >> 16, pos & 0xFFFFu);
		emit_frame(l->d_out, l->out_s16 ? 2u : 0u, j, l->channels, acc);
	}
	free(table);
	return 0;
}

/* for the driver: how many launches of each kind the fake has seen */
unsigned long long crhip_fake_launches(int kind)
{
	return kind >= 0 && kind < 4 ? __atomic_load_n(&g_launches[kind], __ATOMIC_RELAXED) : 0;
}
