/*
 * cr_context.c - process-wide GPU context: error reporting, device selection, plan cache, staging workspace.
 * Host C; reaches the GPU only through the crhip_* shim (crhip.h).  There is deliberately no CPU
 * implementation of the resampling arithmetic anywhere in this library: without a usable device every resample
 * entry point ends in cr_fail().
 */
#include "cr_context.h"

#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/clownresampler_amd.h"

/* tuning variant of the specialised k_poly instances used unless overridden: each instance's measured default */
#define CR_DEFAULT_VARIANT ((int)CRHIP_VARIANT_DEFAULT)

/* ------------------------------------------------------------------------------------------------------- */
/* errors                                                                                                  */
/* ------------------------------------------------------------------------------------------------------- */

static ClownResamplerAMD_ErrorHandler g_handler = NULL;
static void *g_handler_user = NULL;
static __thread int t_last_code = 0;
static __thread unsigned long t_error_serial = 0;
static __thread char t_last_message[512];

int cr_fail(int code, const char *format, ...)
{
	va_list ap;

	va_start(ap, format);
	vsnprintf(t_last_message, sizeof(t_last_message), format, ap);
	va_end(ap);
	t_last_code = code;
	++t_error_serial;

	if (g_handler != NULL)
	{
		g_handler(code, t_last_message, g_handler_user);
	}
	else
	{
		fprintf(stderr, "clownresampler_amd: %s\n", t_last_message);
		fflush(stderr);
		abort();
	}

	return code;
}

unsigned long cr_error_serial(void)
{
	return t_error_serial;
}

int cr_check_hip(int hip_code, const char *what)
{
	if (hip_code != 0)
		cr_fail(CLOWNRESAMPLER_AMD_ERROR_HIP, "%s failed: %s (hipError %d)", what, crhip_error_string(hip_code), hip_code);
	return hip_code;
}

void ClownResamplerAMD_SetErrorHandler(ClownResamplerAMD_ErrorHandler handler, void *user_data)
{
	g_handler = handler;
	g_handler_user = user_data;
}

int ClownResamplerAMD_LastErrorCode(void)
{
	return t_last_code;
}

const char *ClownResamplerAMD_LastErrorMessage(void)
{
	return t_last_code != 0 ? t_last_message : "";
}

void ClownResamplerAMD_ClearError(void)
{
	t_last_code = 0;
	t_last_message[0] = '\0';
}

/* ------------------------------------------------------------------------------------------------------- */
/* device                                                                                                  */
/* ------------------------------------------------------------------------------------------------------- */

static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static int g_device = 0;
static int g_device_ready = 0;
static crhip_device_info g_info;
static ClownResamplerAMD_Plan *g_plans = NULL;
static uint64_t g_plan_clock = 0;
static size_t g_plan_limit = 64;   /* unpinned plans kept; ClownResamplerAMD_SetPlanCacheLimit */
static cr_workspace g_workspace;
#define CR_EXTRA_SETS 2 /* further staging sets of the pipelined host path (cr_run_host): three in all; same lock as g_workspace */
static cr_workspace g_workspace_more[CR_EXTRA_SETS]; /* ([0].stream is the download stream) */
static int g_workspace_busy = 0;
/* small host-buffer calls: one pinned, device-visible block (input at the front, output behind it) that the kernel reads and
   writes across PCIe itself - see cr_run_host */
#define CR_SMALL_CALL_BYTES (128u * 1024u)
static unsigned char *g_small = NULL;
static int g_force_generic = 0;
/* Ticket blocks for k_poly's dynamic tile scheduling (crhip.h CRHIP_TICKET_WORDS counters each), zeroed once; a launch takes
   a block and leaves it zeroed.  Two launches that may run AT THE SAME TIME must not share a block.  Launches on one stream
   never do, so the blocks are handed out per stream: every stream seen gets a ring of CR_RING_SLOTS blocks (a ring rather
   than one block so that the launches of a stream capture - whose graphs may later be replayed side by side - differ too).
   Up to CR_TICKET_RINGS streams are live at once; one more evicts the ring used longest ago, after a device
   synchronise (rare, and the only way to know that ring's launches are through without holding on to its stream). */
#define CR_TICKET_SLOTS 512u
#define CR_TICKET_RINGS 8u
#define CR_RING_SLOTS (CR_TICKET_SLOTS / CR_TICKET_RINGS)
static uint32_t *g_tickets = NULL;
static struct
{
	void *stream;
	int used;
	unsigned next;
	unsigned long long last_use;
} g_rings[CR_TICKET_RINGS];
static unsigned long long g_ring_clock = 0;
static pthread_mutex_t g_ring_lock = PTHREAD_MUTEX_INITIALIZER;

static uint32_t *ticket_block_for(void *stream)
{
	unsigned r, pick = CR_TICKET_RINGS, oldest = 0;
	uint32_t *block;

	pthread_mutex_lock(&g_ring_lock);
	for (r = 0; r < CR_TICKET_RINGS; ++r)
	{
		if (g_rings[r].used && g_rings[r].stream == stream)
		{
			pick = r;
			break;
		}
		if (!g_rings[r].used && pick == CR_TICKET_RINGS)
			pick = r;
		if (g_rings[r].used && g_rings[r].last_use < g_rings[oldest].last_use)
			oldest = r;
	}
	if (pick == CR_TICKET_RINGS)
	{
		/* all rings belong to other streams: take over the one used longest ago, once nothing can be running on it */
		crhip_device_sync();
		pick = oldest;
		g_rings[pick].used = 0;
	}
	if (!g_rings[pick].used)
	{
		g_rings[pick].used = 1;
		g_rings[pick].stream = stream;
		g_rings[pick].next = 0;
	}
	g_rings[pick].last_use = ++g_ring_clock;
	block = g_tickets + CRHIP_TICKET_WORDS * (pick * CR_RING_SLOTS + g_rings[pick].next++ % CR_RING_SLOTS);
	pthread_mutex_unlock(&g_ring_lock);
	return block;
}
static unsigned long long *g_debug_stamps = NULL;
static int g_variant = -1; /* -1: CLOWNRESAMPLER_AMD_VARIANT from the environment, else the default */
static pthread_mutex_t g_workspace_lock = PTHREAD_MUTEX_INITIALIZER;

int ClownResamplerAMD_DeviceCount(void)
{
	int count = 0;

	if (crhip_device_count(&count) != 0)
		return 0;

	return count;
}

static int ensure_device_locked(void)
{
	int count = 0;
	int e;

	if (g_device_ready)
		return cr_check_hip(crhip_set_device(g_device), "hipSetDevice"); /* the current device is per thread */

	e = crhip_device_count(&count);

	if (e != 0 || count <= 0)
		return cr_fail(CLOWNRESAMPLER_AMD_ERROR_NO_DEVICE,
		               "no usable HIP device (%s); this library has no CPU fallback for the resampling path",
		               e != 0 ? crhip_error_string(e) : "device count is 0");

	if (g_device >= count)
		return cr_fail(CLOWNRESAMPLER_AMD_ERROR_NO_DEVICE, "device %d selected but only %d present", g_device, count);

	if (cr_check_hip(crhip_set_device(g_device), "hipSetDevice") != 0)
		return CLOWNRESAMPLER_AMD_ERROR_HIP;

	if (cr_check_hip(crhip_get_device_info(g_device, &g_info), "hipGetDeviceProperties") != 0)
		return CLOWNRESAMPLER_AMD_ERROR_HIP;

	if (g_tickets == NULL)
	{
		if (cr_check_hip(crhip_malloc((void **)&g_tickets, CR_TICKET_SLOTS * CRHIP_TICKET_WORDS * sizeof(uint32_t)), "hipMalloc(tickets)") != 0
		 || cr_check_hip(crhip_memset(g_tickets, 0, CR_TICKET_SLOTS * CRHIP_TICKET_WORDS * sizeof(uint32_t), NULL), "hipMemset(tickets)") != 0
		 || cr_check_hip(crhip_stream_sync(NULL), "hipStreamSynchronize") != 0)
			return CLOWNRESAMPLER_AMD_ERROR_HIP;
	}

	g_device_ready = 1;
	return 0;
}

int cr_ensure_device(void)
{
	int r;

	pthread_mutex_lock(&g_lock);
	r = ensure_device_locked();
	pthread_mutex_unlock(&g_lock);
	return r;
}

const crhip_device_info *cr_device_info(void)
{
	return &g_info;
}

static cr_stream *g_streams = NULL;
static uint64_t g_stream_serial = 0;
static size_t g_stream_max_frames = (size_t)1 << 18;

static void store_release(cr_plan_store *store, int device_usable)
{
	if (store == NULL || --store->refs > 0)
		return;

	if (device_usable)
	{
		/* hipFree waits for the device: a launch that was enqueued with these rows has finished by the time they go */
		crhip_free(store->d_table);
		crhip_free(store->d_rows);
	}
	cr_poly_free(&store->poly);
	free(store);
}

static void release_everything_locked(void)
{
	/* streaming side windows are host memory: they stay valid across device changes and are only dropped by Shutdown */
	ClownResamplerAMD_Plan *p = g_plans;

	while (p != NULL)
	{
		ClownResamplerAMD_Plan *next = p->next;

		store_release(p->store, g_device_ready);
		free(p);
		p = next;
	}
	g_plans = NULL;

	if (g_device_ready)
	{
		crhip_free(g_tickets);
		g_tickets = NULL;
		memset(g_rings, 0, sizeof(g_rings));
		crhip_free(g_workspace.d_in);
		crhip_free(g_workspace.d_out);
		if (g_workspace.stream != NULL)
			crhip_stream_destroy(g_workspace.stream);
		if (g_small != NULL)
			crhip_host_free(g_small);
		{
			int k;
			for (k = 0; k < CR_EXTRA_SETS; ++k)
			{
				crhip_free(g_workspace_more[k].d_in);
				crhip_free(g_workspace_more[k].d_out);
				if (g_workspace_more[k].stream != NULL)
					crhip_stream_destroy(g_workspace_more[k].stream);
			}
		}
	}
	g_small = NULL;
	memset(&g_workspace, 0, sizeof(g_workspace));
	memset(g_workspace_more, 0, sizeof(g_workspace_more));
}

int ClownResamplerAMD_SetDevice(int ordinal)
{
	int r = 0;

	pthread_mutex_lock(&g_lock);
	if (ordinal != g_device || !g_device_ready)
	{
		if (g_device_ready)
			crhip_set_device(g_device);
		release_everything_locked(); /* plans and staging belong to the old device */
		g_device = ordinal;
		g_device_ready = 0;
		r = ensure_device_locked();
	}
	pthread_mutex_unlock(&g_lock);
	return r;
}

int ClownResamplerAMD_GetDevice(void)
{
	return g_device;
}

void ClownResamplerAMD_Shutdown(void)
{
	pthread_mutex_lock(&g_lock);
	while (g_streams != NULL)
	{
		cr_stream *next = g_streams->next;
		free(g_streams->window);
		free(g_streams);
		g_streams = next;
	}
	if (g_device_ready)
		crhip_set_device(g_device);
	release_everything_locked();
	g_device_ready = 0;
	pthread_mutex_unlock(&g_lock);
}

void *ClownResamplerAMD_DeviceAlloc(size_t bytes)
{
	void *p = NULL;

	if (cr_ensure_device() != 0)
		return NULL;
	if (cr_check_hip(crhip_malloc(&p, bytes != 0 ? bytes : 16), "hipMalloc") != 0)
		return NULL;
	return p;
}

void ClownResamplerAMD_DeviceFree(void *device_pointer)
{
	if (device_pointer != NULL && cr_ensure_device() == 0)
		cr_check_hip(crhip_free(device_pointer), "hipFree");
}

int ClownResamplerAMD_CopyToDevice(void *device_destination, const void *host_source, size_t bytes)
{
	if (cr_ensure_device() != 0)
		return -1;
	if (cr_check_hip(crhip_memcpy_h2d(device_destination, host_source, bytes, NULL), "hipMemcpyAsync(H2D)") != 0)
		return -1;
	return cr_check_hip(crhip_stream_sync(NULL), "hipStreamSynchronize") != 0 ? -1 : 0;
}

int ClownResamplerAMD_CopyFromDevice(void *host_destination, const void *device_source, size_t bytes)
{
	if (cr_ensure_device() != 0)
		return -1;
	if (cr_check_hip(crhip_memcpy_d2h(host_destination, device_source, bytes, NULL), "hipMemcpyAsync(D2H)") != 0)
		return -1;
	return cr_check_hip(crhip_stream_sync(NULL), "hipStreamSynchronize") != 0 ? -1 : 0;
}

int ClownResamplerAMD_StreamSynchronize(void *hip_stream)
{
	if (cr_ensure_device() != 0)
		return -1;
	return cr_check_hip(crhip_stream_sync(hip_stream), "hipStreamSynchronize") != 0 ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------------------- */
/* plans                                                                                                   */
/* ------------------------------------------------------------------------------------------------------- */

static void fill_poly_launch(const ClownResamplerAMD_Plan *plan, crhip_poly_launch *l);

static uint32_t current_variant(void)
{
	if (g_variant < 0)
	{
		const char *e = getenv("CLOWNRESAMPLER_AMD_VARIANT");
		g_variant = (e != NULL && *e != '\0') ? atoi(e) : CR_DEFAULT_VARIANT;
		if (g_variant < 0 || (g_variant >= crhip_poly_variants() && !(g_variant >= 1000 && g_variant < 1010)))
			g_variant = CR_DEFAULT_VARIANT;
	}
	return (uint32_t)g_variant;
}

static uint32_t supported_poly_channels(uint32_t channels)
{
	/* 1..8 one lane per frame; 9..16 two lanes per frame (cr_kernels.hip runtime_split; odd counts with a phantom channel) */
	return channels >= 1 && channels <= 16;
}

static uint32_t plan_image_stride(const ClownResamplerAMD_Plan *plan)
{
	const int special = crhip_poly_has_instance(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode)
	                    && getenv("CLOWNRESAMPLER_AMD_NO_SPECIAL") == NULL;
	return special ? plan->poly.row_stride : 4u * ((plan->poly.slots + 3u) / 4u + 1u);
}

/* Launch geometry of k_poly for this plan on the current device. */
static void plan_geometry(ClownResamplerAMD_Plan *plan)
{
	const uint32_t frame_bytes = plan->channels * 2u;
	/* int32 per row of the device image: see cr_poly_device_image */
	const uint32_t image_stride = plan_image_stride(plan);
	const uint32_t rows_bytes = cr_poly_plane_rows(&plan->poly) * image_stride * 4u;
	uint32_t tile_bytes, cap_frames, per_cu;
	uint64_t tile;

	uint32_t frames_multiple = 0;
	/* frames of one tap window as a tile has to hold it: the slots plus the largest shift of a phase's window (shifted rows) */
	const uint32_t window_slots = plan->poly.slots + plan->poly.window_extra;

	plan->specialised = (uint32_t)crhip_poly_has_instance(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode);
	if (getenv("CLOWNRESAMPLER_AMD_NO_SPECIAL") != NULL) /* tuning hook: time the run-time-slot instance instead */
		plan->specialised = 0;
	crhip_poly_geometry(plan->channels, plan->specialised ? plan->poly.slots : 0xFFFFu, plan->poly.row_mode, plan->poly.norm_mode, plan->variant, &plan->threads, &plan->vecs, &frames_multiple);
	if (plan->variant == 28u || plan->variant == 29u || (plan->variant == 0xFFFFu && crhip_poly_default_is_mad(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode)))
	{
		/* k_poly with the 64-bit multiply-add chain: compiled for fixed weight signs per slot, like k_up */
		uint32_t negmask = 0, pos_bits = 0, neg_bits = 0;
		int ok = crhip_poly_up_negmask(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode, &negmask);

		if (ok)
		{
			cr_poly_slot_signs(&plan->poly, &pos_bits, &neg_bits);
			ok = (neg_bits & ~negmask) == 0 && (pos_bits & negmask) == 0;
		}

		if (!ok)
		{
			plan->variant = crhip_poly_fallback_variant();
			crhip_poly_geometry(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode, plan->variant, &plan->threads, &plan->vecs, &frames_multiple);
		}
	}

	if (plan->vecs >= 200u)
	{
		/* k_up: a lane owns one input position and produces all of its output frames; a wave-tile is as many output
		   frames as 64 positions are sure to cover, staged through LDS for coalesced stores.  Qualifies when there are
		   at least two frames per position, the weight signs per slot are the ones the instance was compiled for, and
		   rows + staging fit the LDS. */
		uint32_t negmask = 0, pos_bits = 0, neg_bits = 0;
		const uint32_t unit = plan->channels * 4u;
		uint64_t wave_tile = ((uint64_t)63u << 16) / plan->increment + 1u;   /* 65535 + (wave_tile - 1) * increment < 64 * 65536 */
		int ok = plan->increment <= 32768u && plan->increment >= 4096u
		      && crhip_poly_up_negmask(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode, &negmask);

		if (ok)
		{
			cr_poly_slot_signs(&plan->poly, &pos_bits, &neg_bits);
			ok = (neg_bits & ~negmask) == 0 && (pos_bits & negmask) == 0;
		}

		if (wave_tile > frames_multiple)
			wave_tile = frames_multiple;
		{
			/* ... and no larger than the staging space one workgroup per CU leaves each wave */
			const uint32_t waves = plan->threads / 64u;
			const uint32_t fixed = rows_bytes + 16u + waves * 2u * 1024u;
			const uint32_t lds = (uint32_t)g_info.max_lds_per_block < 160u * 1024u ? (uint32_t)g_info.max_lds_per_block : 160u * 1024u;
			const uint64_t room = lds > fixed ? ((lds - fixed) / waves & ~15u) / unit : 0;

			if (wave_tile > room)
				wave_tile = room;
			if (wave_tile < 64u)
				ok = 0;
		}

		if (ok)
		{
			const uint32_t stage_bytes = ((uint32_t)wave_tile * unit + 15u) & ~15u;

			plan->lds_bytes = rows_bytes + (plan->threads / 64u) * (2u * 1024u + stage_bytes) + 16u;
			plan->tile_frames = (uint32_t)wave_tile * 4u;
			per_cu = (160u * 1024u) / plan->lds_bytes;
			if (per_cu > 2048u / plan->threads)
				per_cu = 2048u / plan->threads;
			if (per_cu >= 1u && plan->lds_bytes <= (uint32_t)g_info.max_lds_per_block)
			{
				plan->max_blocks = per_cu * (uint32_t)(g_info.compute_units > 0 ? g_info.compute_units : 256);
				return;
			}
		}

		if (getenv("CLOWNRESAMPLER_AMD_DEBUG") != NULL)
			fprintf(stderr, "clownresampler_amd: k_up not used: increment %llu, ok %d, signs +%#x -%#x against mask %#x, wave tile %llu, lds %u of %d\n",
			        (unsigned long long)plan->increment, ok, pos_bits, neg_bits, negmask, (unsigned long long)wave_tile, plan->lds_bytes, g_info.max_lds_per_block);
		plan->variant = crhip_poly_up_fallback_variant(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode);
		crhip_poly_geometry(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode, plan->variant, &plan->threads, &plan->vecs, &frames_multiple);
	}

	if (plan->vecs >= 100u)
	{
		/* k_wave: every wave streams wave-tiles of 64 * 4 frames through a private, double-buffered (vecs - 100) KiB
		   slice of LDS; work is handed out in chunks of 4 wave-tiles (frames_multiple) */
		const uint32_t piece_bytes = (plan->vecs - 100u) * 1024u;
		const uint32_t wave_tile = frames_multiple / 4u;
		const uint64_t last_rel = (65535u + (uint64_t)(wave_tile - 1u) * plan->increment) >> 16;
		const uint64_t window = 12u + (last_rel + window_slots) * frame_bytes;

		if (window <= piece_bytes && (uint64_t)wave_tile * plan->increment < (1ull << 32) - 65536u)
		{
			plan->lds_bytes = rows_bytes + (plan->threads / 64u) * 2u * piece_bytes + 16u; /* + the retired-waves counter */
			plan->tile_frames = frames_multiple;
			per_cu = (160u * 1024u) / plan->lds_bytes;
			if (per_cu > 2048u / plan->threads)
				per_cu = 2048u / plan->threads;
			if (per_cu >= 1u && plan->lds_bytes <= (uint32_t)g_info.max_lds_per_block)
			{
				plan->max_blocks = per_cu * (uint32_t)(g_info.compute_units > 0 ? g_info.compute_units : 256);
				return;
			}
		}

		/* the window of this configuration does not fit a wave's slice: use a k_poly variant instead */
		plan->variant = crhip_poly_fallback_variant();
		crhip_poly_geometry(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode, plan->variant, &plan->threads, &plan->vecs, &frames_multiple);
	}

	tile_bytes = plan->vecs * 16u * plan->threads;
	plan->lds_bytes = rows_bytes + 2u * tile_bytes + 16u; /* + the ticket mailbox */

	if (plan->lds_bytes > (uint32_t)g_info.max_lds_per_block)
	{
		plan->use_poly = 0;
		plan->generic_reason = "polyphase rows + tiles exceed the LDS of one workgroup";
		return;
	}

	/* frames the tile image can hold after the (< 16 byte) alignment shift */
	cap_frames = (tile_bytes - 16u) / frame_bytes;

	if (cap_frames <= window_slots)
	{
		plan->use_poly = 0;
		plan->generic_reason = "tap window longer than an LDS tile";
		return;
	}

	/* largest tile with ((65535 + (tile - 1) * increment) >> 16) + slots <= cap_frames */
	tile = ((uint64_t)(cap_frames - window_slots) << 16) / plan->increment + 1u;

	/* 32-bit relative positions and the 24-bit multiplier of the kernel */
	if (tile > ((1ull << 32) - 65536u) / plan->increment)
		tile = ((1ull << 32) - 65536u) / plan->increment;
	if (tile > (1u << 24) - 1u)
		tile = (1u << 24) - 1u;
	/* whole groups of threads * frames-in-flight; 4, 2 or 1 groups per tile run as straight-line code in the kernel */
	{
		const char *e = getenv("CLOWNRESAMPLER_AMD_TILE_GROUPS"); /* tuning hook: cap the groups per tile */
		/* 13 and 15 channels (7-8 channels per lane plus the phantom channel): one group per tile.  With four groups - strong
		   upsampling - the straight-line tile measured 0.18 of the roofline against 0.30 (profiles/r01_channel_table.log);
		   at 44.1 <-> 48 kHz, where two groups fit, one costs 1-2 % */
		const uint64_t wide_phantom = (!plan->specialised && plan->channels > 12u && plan->channels % 2u == 1u) ? 1u : 0u;
		const uint64_t cap = (e != NULL && atoi(e) > 0) ? (uint64_t)atoi(e) * frames_multiple : wide_phantom * frames_multiple;
		if (cap != 0 && tile > cap)
			tile = cap;
	}
	if (tile >= 4u * frames_multiple)
		tile = 4u * frames_multiple;
	else if (tile >= 2u * frames_multiple)
		tile = 2u * frames_multiple;
	else if (tile >= frames_multiple)
		tile = frames_multiple;

	if (tile == 0)
	{
		plan->use_poly = 0;
		plan->generic_reason = "increment too large for an LDS tile";
		return;
	}

	plan->tile_frames = (uint32_t)tile;

	/* persistent grid: as many workgroups as the LDS footprint lets the chip hold at once */
	per_cu = (160u * 1024u) / plan->lds_bytes;
	if (per_cu > 2048u / plan->threads)
		per_cu = 2048u / plan->threads;
	if (per_cu < 1u)
		per_cu = 1u;
	plan->max_blocks = per_cu * (uint32_t)(g_info.compute_units > 0 ? g_info.compute_units : 256);
}

static int plan_key_matches(const ClownResamplerAMD_Plan *plan, uint64_t table_hash, unsigned radius, const cr_config *cfg, uint32_t channels)
{
	return plan->table_hash == table_hash && plan->radius == radius && plan->channels == channels
	    && plan->key_variant == current_variant() && plan->device == g_device && memcmp(&plan->cfg, cfg, sizeof(*cfg)) == 0;
}

/* Drops unpinned, unheld plans, least recently used first, until at most g_plan_limit unpinned plans remain. */
static void evict_plans_locked(void)
{
	for (;;)
	{
		ClownResamplerAMD_Plan **link, **victim = NULL;
		size_t unpinned = 0;

		for (link = &g_plans; *link != NULL; link = &(*link)->next)
		{
			if ((*link)->pinned)
				continue;
			++unpinned;
			if ((*link)->users == 0 && (victim == NULL || (*link)->last_use < (*victim)->last_use))
				victim = link;
		}

		if (unpinned <= g_plan_limit || victim == NULL)
			return;

		{
			ClownResamplerAMD_Plan *plan = *victim;
			*victim = plan->next;
			store_release(plan->store, g_device_ready);
			free(plan);
		}
	}
}

void ClownResamplerAMD_SetPlanCacheLimit(size_t plans)
{
	pthread_mutex_lock(&g_lock);
	g_plan_limit = plans;
	evict_plans_locked();
	pthread_mutex_unlock(&g_lock);
}

size_t ClownResamplerAMD_PlanCacheCount(void)
{
	const ClownResamplerAMD_Plan *plan;
	size_t n = 0;

	pthread_mutex_lock(&g_lock);
	for (plan = g_plans; plan != NULL; plan = plan->next)
		++n;
	pthread_mutex_unlock(&g_lock);
	return n;
}

void cr_plan_release(const ClownResamplerAMD_Plan *plan_in)
{
	ClownResamplerAMD_Plan *plan = (ClownResamplerAMD_Plan *)plan_in;

	if (plan == NULL)
		return;

	pthread_mutex_lock(&g_lock);
	if (plan->users != 0)
		--plan->users;
	evict_plans_locked();
	pthread_mutex_unlock(&g_lock);
}

ClownResamplerAMD_Plan *cr_plan_get(uint64_t table_hash, size_t table_len, cr_table_fill fill_table, const void *user,
                                    unsigned radius, const cr_config *cfg, uint32_t channels, uint64_t increment, int pin)
{
	ClownResamplerAMD_Plan *plan, *sibling = NULL;
	cr_plan_store *store = NULL;
	int32_t *table = NULL;

	pthread_mutex_lock(&g_lock);

	if (ensure_device_locked() != 0)
		goto fail;

	for (plan = g_plans; plan != NULL; plan = plan->next)
	{
		if (!plan_key_matches(plan, table_hash, radius, cfg, channels))
			continue;

		if (plan->increment == increment)
		{
			plan->last_use = ++g_plan_clock;
			plan->users += 1;
			plan->pinned |= pin;
			pthread_mutex_unlock(&g_lock);
			return plan;
		}

		sibling = plan; /* same rows, different increment */
	}

	if (channels == 0 || channels > CRHIP_MAX_CHANNELS)
	{
		cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "channel count %u outside 1..%d (CLOWNRESAMPLER_MAXIMUM_CHANNELS)", channels, CRHIP_MAX_CHANNELS);
		goto fail;
	}

	if (increment == 0 || increment >= 0xFFFFFFFFull)
	{
		cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "increment %llu is not a valid 16.16 ratio (state not initialised?)", (unsigned long long)increment);
		goto fail;
	}

	plan = (ClownResamplerAMD_Plan *)calloc(1, sizeof(*plan));
	if (plan == NULL)
	{
		cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "out of host memory");
		goto fail;
	}

	plan->table_hash = table_hash;
	plan->radius = radius;
	plan->cfg = *cfg;
	plan->channels = channels;
	plan->increment = increment;
	plan->device = g_device;
	plan->key_variant = current_variant();
	plan->variant = plan->key_variant;
	plan->table_len = (uint32_t)table_len;

	if (sibling != NULL)
	{
		store = sibling->store;
		store->refs += 1;
	}
	else
	{
		table = (int32_t *)malloc(table_len * sizeof(int32_t));
		store = (cr_plan_store *)calloc(1, sizeof(*store));

		if (table == NULL || store == NULL)
		{
			free(store);
			free(plan);
			cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "out of host memory");
			goto fail;
		}

		store->refs = 1;
		store->table_len = (uint32_t)table_len;
		store->rows_layout = -1;

		if (fill_table(user, table, table_len) != 0)
		{
			free(store);
			free(plan);
			cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "a Lanczos table entry does not fit 32 bits");
			goto fail;
		}

		if (cr_poly_build(table, table_len, cfg, &store->poly) != 0)
		{
			/* the reference itself would trap or read outside its table with this configuration */
			cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "unusable configuration: %s", store->poly.reason);
			free(store);
			free(plan);
			goto fail;
		}

		if (cr_check_hip(crhip_malloc((void **)&store->d_table, table_len * sizeof(int32_t)), "hipMalloc(table)") != 0
		 || cr_check_hip(crhip_memcpy_h2d(store->d_table, table, table_len * sizeof(int32_t), NULL), "hipMemcpy(table)") != 0
		 || cr_check_hip(crhip_stream_sync(NULL), "hipStreamSynchronize") != 0)
			goto fail_plan;
	}

	plan->store = store;
	plan->poly = store->poly;       /* view: the arrays belong to the store */
	plan->d_table = store->d_table;

	plan->use_poly = plan->poly.eligible && plan->poly.weights != NULL;
	plan->generic_reason = plan->poly.reason;

	if (plan->use_poly && !supported_poly_channels(channels))
	{
		plan->use_poly = 0;
		plan->generic_reason = "no polyphase kernel instance for this channel count";
	}

	if (plan->use_poly && increment >= (1u << 24))
	{
		plan->use_poly = 0;
		plan->generic_reason = "increment does not fit the 24-bit multiplier";
	}

	if (plan->use_poly)
		plan_geometry(plan);

	if (plan->use_poly)
	{
		const int layout = plan->specialised ? CR_IMAGE_COMPACT : CR_IMAGE_SPLIT;

		/* the bank-conflict model walks a few thousand wave footprints (~0.4 ms): only worth running for an instance that
		   can read a swizzled image, and none is built at present */
		plan->swizzle = 0;
		if (crhip_poly_swizzled(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode, plan->variant))
			plan->swizzle = cr_poly_pick_swizzle(&plan->poly, increment, &plan->conflict_plain, &plan->conflict_best);

		if (store->d_rows != NULL && (store->rows_layout != layout || store->swizzle != plan->swizzle))
		{
			/* cannot happen while the layout is a function of (channels, slots, row mode, normalisation), which siblings
			   share; if a tuning hook ever breaks that, the generic kernel is still right */
			plan->use_poly = 0;
			plan->generic_reason = "plans of one configuration disagree on the layout of their rows";
		}
		else if (store->d_rows == NULL)
		{
			int32_t *image = cr_poly_device_image(&store->poly, plan->swizzle, layout, &store->device_row_stride);
			const size_t bytes = (size_t)cr_poly_plane_rows(&store->poly) * store->device_row_stride * sizeof(int32_t);

			if (image == NULL)
			{
				cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "out of host memory");
				goto fail_plan;
			}

			if (cr_check_hip(crhip_malloc((void **)&store->d_rows, bytes), "hipMalloc(rows)") != 0
			 || cr_check_hip(crhip_memcpy_h2d(store->d_rows, image, bytes, NULL), "hipMemcpy(rows)") != 0
			 || cr_check_hip(crhip_stream_sync(NULL), "hipStreamSynchronize") != 0)
			{
				free(image);
				goto fail_plan;
			}

			free(image);
			store->rows_layout = layout;
			store->swizzle = plan->swizzle;
			store->plane_rows = cr_poly_plane_rows(&store->poly);
		}

	}

	if (plan->use_poly)
	{
		plan->d_rows = store->d_rows;
		plan->plane_rows = store->plane_rows;
		plan->device_row_stride = store->device_row_stride;

		{
			/* once per plan, never inside a caller's stream capture */
			crhip_poly_launch l;
			fill_poly_launch(plan, &l);
			if (cr_check_hip(crhip_poly_prepare(&l), "hipFuncSetAttribute(dynamic LDS)") != 0)
				goto fail_plan;
			l.out_s16 = 1; /* the int16-output instance is a different function */
			if (cr_check_hip(crhip_poly_prepare(&l), "hipFuncSetAttribute(dynamic LDS, int16 form)") != 0)
				goto fail_plan;

			/* The persistent grid must not be larger than what is resident at once: workgroups that start after the first
			   batch has left find their statically dealt tiles still waiting (twice the time) or no tickets (harmless).
			   LDS and thread count were accounted for above; registers are the runtime's to know - e.g. above 96 SGPRs a
			   SIMD holds 7 waves, not 8, and a 1024-thread workgroup then has the CU to itself. */
			{
				int form;
				plan->max_blocks_s16 = plan->max_blocks;
				for (form = 0; form < 2; ++form)
				{
					int per_cu = 0, vgprs = 0, static_lds = 0;
					l.out_s16 = (uint32_t)form;
					if (crhip_poly_occupancy(&l, &per_cu, &vgprs, &static_lds) == 0 && per_cu >= 1)
					{
						const uint32_t resident = (uint32_t)per_cu * (uint32_t)(g_info.compute_units > 0 ? g_info.compute_units : 256);
						if (getenv("CLOWNRESAMPLER_AMD_DEBUG") != NULL)
							fprintf(stderr, "clownresampler_amd: plan variant %u (%s output): %u threads, %u B dynamic LDS, %d VGPRs: %d workgroups per CU, grid cap %u -> %u\n",
							        plan->variant, form ? "int16" : "int32", plan->threads, plan->lds_bytes, vgprs, per_cu, form ? plan->max_blocks_s16 : plan->max_blocks, resident);
						if (getenv("CLOWNRESAMPLER_AMD_NO_OCCUPANCY_CLAMP") != NULL) /* (tuning hook: measure without) */
							continue;
						if (form == 0 && resident < plan->max_blocks)
							plan->max_blocks = resident;
						if (form == 1 && resident < plan->max_blocks_s16)
							plan->max_blocks_s16 = resident;
					}
				}
			}
		}
	}

	free(table);
	plan->pinned = pin;
	plan->users = 1;
	plan->last_use = ++g_plan_clock;
	plan->next = g_plans;
	g_plans = plan;
	evict_plans_locked();
	pthread_mutex_unlock(&g_lock);
	return plan;

fail_plan:
	store_release(store, 1);
	free(plan);
fail:
	free(table);
	pthread_mutex_unlock(&g_lock);
	return NULL;
}

/* the launch-independent part of a k_poly launch description */
static void fill_poly_launch(const ClownResamplerAMD_Plan *plan, crhip_poly_launch *l)
{
	memset(l, 0, sizeof(*l));
	l->d_rows = plan->d_rows;
	l->increment = (uint32_t)plan->increment;
	l->channels = plan->channels;
	l->slots = plan->poly.slots;
	l->first_mr = plan->poly.first_mr;
	l->window_extra = plan->poly.window_extra;
	l->first_slot = plan->poly.first_slot;
	l->rows = plan->poly.rows;
	l->row_stride = plan->device_row_stride;
	l->row_mode = plan->poly.row_mode;
	l->norm_mode = plan->poly.norm_mode;
	l->delta = plan->poly.delta;
	l->skr = plan->poly.skr;
	l->step = plan->poly.step;
	l->aff_a = plan->poly.aff_a;
	l->aff_b = plan->poly.aff_b;
	l->aff_c = plan->poly.aff_c;
	l->threads = plan->threads;
	l->vecs = plan->vecs;
	l->tile_frames = plan->tile_frames;
	l->lds_bytes = plan->lds_bytes;
	l->specialised = plan->specialised;
	l->variant = plan->variant;
	l->plane_rows = plan->plane_rows;
	l->swizzle = plan->swizzle;
	l->debug_stamps = g_debug_stamps;
}

int cr_plan_launch(const ClownResamplerAMD_Plan *plan, const void *d_in, uint64_t in_valid_bytes, void *d_out,
                   uint64_t pos_int, uint64_t pos_frac, uint64_t n_out, void *stream, int out_s16)
{
	if (n_out == 0)
		return 0;

	if (plan->use_poly && !g_force_generic && pos_int < (1ull << 47) && n_out < (1ull << 40))
	{
		crhip_poly_launch l;
		uint64_t blocks;

		fill_poly_launch(plan, &l);
		l.d_in = d_in;
		l.in_valid_bytes = in_valid_bytes;
		l.d_out = d_out;
		l.pos0 = (pos_int << 16) + pos_frac;
		l.n_out = n_out;
		l.out_s16 = out_s16 ? 1u : 0u;

		/* tiles are dealt round-robin to a persistent grid (see k_poly) */
		blocks = (n_out + plan->tile_frames - 1) / plan->tile_frames;
		if (plan->vecs >= 100u)
			blocks = (blocks + plan->threads / 64u - 1) / (plan->threads / 64u); /* k_wave hands chunks to WAVES */
		if (blocks > (out_s16 ? plan->max_blocks_s16 : plan->max_blocks))
			blocks = out_s16 ? plan->max_blocks_s16 : plan->max_blocks;
		l.blocks = (uint32_t)blocks;
		{
			/* tickets pay off where workgroups drift apart over many medium-sized tiles; measured on MI355X (profiles/):
			   stereo 3-lobe upsampling gains ~4 %, 8-channel and 8-lobe instances lose 1-8 %: a per-instance default */
			const char *e = getenv("CLOWNRESAMPLER_AMD_DYNAMIC_TILES");
			l.dynamic_tiles = e != NULL ? (uint32_t)(atoi(e) != 0)
			                            : (uint32_t)crhip_poly_dynamic_default(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode);
		}
		l.d_tickets = ticket_block_for(stream);

		return cr_check_hip(crhip_launch_poly(&l, stream), "k_poly launch");
	}
	else
	{
		crhip_generic_launch g;

		memset(&g, 0, sizeof(g));
		g.d_in = d_in;
		g.d_out = d_out;
		g.d_table = plan->d_table;
		g.d_acc_in = NULL;
		g.pos_int = pos_int;
		g.pos_frac = pos_frac;
		g.increment = plan->increment;
		g.n_out = n_out;
		g.skr = plan->cfg.skr;
		g.radius_frames = plan->cfg.radius_frames;
		g.delta = plan->cfg.delta;
		g.step = plan->cfg.step;
		g.table_len = plan->table_len;
		g.channels = plan->channels;
		g.out64 = out_s16 ? 2u : 0u;

		return cr_check_hip(crhip_launch_generic(&g, stream), "k_generic launch");
	}
}

/* ------------------------------------------------------------------------------------------------------- */
/* staging workspace + host-buffer runs                                                                    */
/* ------------------------------------------------------------------------------------------------------- */

static int grow(unsigned char **p, size_t *have, size_t want)
{
	if (*have >= want)
		return 0;

	/* grow-only, with headroom so a stream of slightly different call sizes does not reallocate every time */
	want += want / 4 + 4096;

	if (*p != NULL)
		crhip_free(*p);
	*p = NULL;
	*have = 0;

	if (cr_check_hip(crhip_malloc((void **)p, want), "hipMalloc(staging)") != 0)
		return -1;

	*have = want;
	return 0;
}

cr_workspace *cr_workspace_acquire(size_t in_bytes, size_t out_bytes)
{
	if (cr_ensure_device() != 0)
		return NULL;

	pthread_mutex_lock(&g_workspace_lock);
	g_workspace_busy = 1;

	if (g_workspace.stream == NULL && cr_check_hip(crhip_stream_create(&g_workspace.stream), "hipStreamCreate") != 0)
		goto fail;

	if (grow(&g_workspace.d_in, &g_workspace.d_in_bytes, in_bytes + 64) != 0 || grow(&g_workspace.d_out, &g_workspace.d_out_bytes, out_bytes + 64) != 0)
		goto fail;

	return &g_workspace;

fail:
	g_workspace_busy = 0;
	pthread_mutex_unlock(&g_workspace_lock);
	return NULL;
}

void cr_workspace_release(cr_workspace *ws)
{
	(void)ws;
	g_workspace_busy = 0;
	pthread_mutex_unlock(&g_workspace_lock);
}

/* ---- pipelined host path ----
   hipMemcpyAsync on pageable memory does not return before the copy is done (measured: tools/microbench/pinbench.hip), so
   one host thread can never have an upload and a download in flight together.  For calls of more than one batch a
   second thread therefore does the downloads: while it copies batch b's output (and waits for b's kernel before that),
   the calling thread uploads batch b + 1 into another staging set and launches it.  PCIe is full duplex: the upload
   hides behind the download (pinbench: 4.5-4.8 ms instead of 6.0 ms for the 106 MB + 230 MB of cfg 2; this path:
   6.3 -> 5.0 ms).  Three staging sets: with two, the upload of batch b + 2 had to wait for download b and then ran
   side by side with download b + 1, slowing both, and the download thread idled in between (5.6 ms). */
typedef struct cr_download
{
	pthread_mutex_t lock;
	pthread_cond_t changed;
	struct
	{
		void *host_dst;
		const void *dev_src;
		size_t bytes;
		void *ready;  /* event recorded behind the batch's kernel on the upload/compute stream */
		int queued; /* 1 from submission until the copy has been synchronised */
	} slot[1 + CR_EXTRA_SETS];
	void *stream;     /* the download stream: every D2H of the call, nothing else */
	uint64_t submitted, completed; /* batches; batch b uses slot (and staging set) b % (1 + CR_EXTRA_SETS) */
	int quit;
	int hip_error; /* first failing HIP call of the thread (reported by the caller's thread, whose error state the API exposes) */
} cr_download;

static void *download_thread(void *arg)
{
	cr_download *d = (cr_download *)arg;
	int code = crhip_set_device(g_device);

	pthread_mutex_lock(&d->lock);
	if (code != 0 && d->hip_error == 0)
		d->hip_error = code;

	for (;;)
	{
		const unsigned s = (unsigned)(d->completed % (1u + CR_EXTRA_SETS));

		while (d->completed == d->submitted && !d->quit)
			pthread_cond_wait(&d->changed, &d->lock);
		if (d->completed == d->submitted)
			break; /* quit and nothing left */

		pthread_mutex_unlock(&d->lock);
		/* stream order: after the batch's kernel */
		code = crhip_stream_wait_event(d->stream, d->slot[s].ready);
		if (code == 0)
			code = crhip_memcpy_d2h(d->slot[s].host_dst, d->slot[s].dev_src, d->slot[s].bytes, d->stream);
		if (code == 0)
			code = crhip_stream_sync(d->stream);
		pthread_mutex_lock(&d->lock);

		if (code != 0 && d->hip_error == 0)
			d->hip_error = code;
		d->slot[s].queued = 0;
		++d->completed;
		pthread_cond_broadcast(&d->changed);
	}

	pthread_mutex_unlock(&d->lock);
	return NULL;
}

int cr_run_host(const ClownResamplerAMD_Plan *plan, const int16_t *host_in, uint64_t in_frames, uint64_t pos_int,
                uint64_t pos_frac, uint64_t n_out, void *host_out, int out_s16)
{
	/* bounded batches keep the staging buffers small whatever the stream length */
	const uint64_t batch_frames = 4u << 20;
	const size_t frame_in = (size_t)plan->channels * sizeof(int16_t);
	const size_t frame_out = (size_t)plan->channels * (out_s16 ? sizeof(int16_t) : sizeof(int32_t));
	const int pipelined = n_out > batch_frames && getenv("CLOWNRESAMPLER_AMD_NO_HOST_PIPELINE") == NULL;
	uint64_t done = 0, batch = 0;
	cr_download dl;
	pthread_t thread;
	int have_thread = 0, bad = 0;
	cr_workspace *ws;

	if (n_out == 0)
		return 0;

	/* SMALL calls (a sound-card sized request, a refill of the streaming API): two hipMemcpyAsync and a launch cost ~35 us
	   whatever the size, most of it in the two copies.  Up to CR_SMALL_CALL_BYTES the kernel therefore works on pinned host
	   memory directly - the input is copied into it by the CPU, the LDS-DMA reads it and the stores write the result across
	   PCIe, and one synchronise later the CPU copies the result out: one launch, no copy calls (480 frames: 34 -> ~15 us). */
	if (n_out <= batch_frames && getenv("CLOWNRESAMPLER_AMD_NO_SMALL_CALL_PATH") == NULL)
	{
		uint64_t pi = pos_int, extent = cr_input_extent(&plan->cfg, 0, pos_frac, plan->increment, n_out);
		size_t in_bytes, out_bytes, out_at;

		if (pi >= in_frames)
			extent = 0;
		else if (extent > in_frames - pi)
			extent = in_frames - pi;
		in_bytes = (size_t)extent * frame_in;
		out_bytes = (size_t)n_out * frame_out;
		out_at = (in_bytes + 255u) & ~(size_t)255u;

		if (out_at + out_bytes + 64u <= CR_SMALL_CALL_BYTES)
		{
			ws = cr_workspace_acquire(0, 0); /* (the lock and the stream) */
			if (ws == NULL)
				return -1;
			if (g_small == NULL && cr_check_hip(crhip_host_alloc((void **)&g_small, CR_SMALL_CALL_BYTES), "hipHostMalloc(small-call block)") != 0)
			{
				g_small = NULL;
				cr_workspace_release(ws);
				return -1;
			}
			memcpy(g_small, host_in + pi * plan->channels, in_bytes);
			bad = cr_plan_launch(plan, g_small, in_bytes, g_small + out_at, 0, pos_frac, n_out, ws->stream, out_s16) != 0
			   || cr_check_hip(crhip_stream_sync(ws->stream), "hipStreamSynchronize") != 0;
			if (!bad)
				memcpy(host_out, g_small + out_at, out_bytes);
			cr_workspace_release(ws);
			return bad ? -1 : 0;
		}
	}

	/* both staging sets, sized for a full batch, under the one workspace lock for the whole call */
	{
		const uint64_t n0 = n_out < batch_frames ? n_out : batch_frames;
		uint64_t extent0 = cr_input_extent(&plan->cfg, 0, 65535u, plan->increment, n0);

		if (extent0 > in_frames)
			extent0 = in_frames;
		ws = cr_workspace_acquire((size_t)extent0 * frame_in, (size_t)n0 * frame_out);
		if (ws == NULL)
			return -1;

		if (pipelined)
		{
			int k, failed = g_workspace_more[0].stream == NULL && cr_check_hip(crhip_stream_create(&g_workspace_more[0].stream), "hipStreamCreate") != 0;

			for (k = 0; k < CR_EXTRA_SETS && !failed; ++k)
				failed = grow(&g_workspace_more[k].d_in, &g_workspace_more[k].d_in_bytes, (size_t)extent0 * frame_in + 64) != 0
				      || grow(&g_workspace_more[k].d_out, &g_workspace_more[k].d_out_bytes, (size_t)n0 * frame_out + 64) != 0;
			if (failed)
			{
				cr_workspace_release(ws);
				return -1;
			}

			memset(&dl, 0, sizeof(dl));
			pthread_mutex_init(&dl.lock, NULL);
			pthread_cond_init(&dl.changed, NULL);
			/* uploads and kernels all go to ONE stream (ws->stream), downloads all to another (the second set's): copies
			   of one direction per stream is what lets the runtime run the two directions side by side (measured: with
			   each batch's three steps on its own stream the download stalled for as long as the next upload ran) */
			dl.stream = g_workspace_more[0].stream;
			for (k = 0; k < 1 + CR_EXTRA_SETS && !failed; ++k)
				failed = cr_check_hip(crhip_event_create(&dl.slot[k].ready), "hipEventCreate") != 0;
			if (failed)
			{
				for (k = 0; k < 1 + CR_EXTRA_SETS; ++k)
					if (dl.slot[k].ready != NULL)
						crhip_event_destroy(dl.slot[k].ready);
				cr_workspace_release(ws);
				return -1;
			}
			have_thread = pthread_create(&thread, NULL, download_thread, &dl) == 0;
			/* (no thread: the loop below degrades to one batch at a time) */
		}
	}

	while (done < n_out && !bad)
	{
		const uint64_t n = n_out - done < batch_frames ? n_out - done : batch_frames;
		uint64_t pi = pos_int, pf = pos_frac, extent;
		const unsigned set = have_thread ? (unsigned)(batch % (1u + CR_EXTRA_SETS)) : 0u;
		cr_workspace *w = set == 0u ? ws : &g_workspace_more[set - 1u];

		cr_advance(&pi, &pf, plan->increment, done);

		/* padded-buffer frames [pi, pi + extent) cover everything this batch reads (clownresampler.h:995-996) */
		extent = cr_input_extent(&plan->cfg, 0, pf, plan->increment, n);
		if (pi >= in_frames)
			extent = 0;
		else if (extent > in_frames - pi)
			extent = in_frames - pi;

		if (grow(&w->d_in, &w->d_in_bytes, (size_t)extent * frame_in + 64) != 0 || grow(&w->d_out, &w->d_out_bytes, (size_t)n * frame_out + 64) != 0)
		{
			bad = 1; /* (sized for a full batch above: only reached if that estimate was short) */
			break;
		}

		if (have_thread)
		{
			/* this staging set was last used three batches ago: its download must be through */
			pthread_mutex_lock(&dl.lock);
			while (dl.slot[set].queued)
				pthread_cond_wait(&dl.changed, &dl.lock);
			bad = dl.hip_error != 0;
			pthread_mutex_unlock(&dl.lock);
			if (bad)
				break;
		}

		bad = cr_check_hip(crhip_memcpy_h2d(w->d_in, host_in + pi * plan->channels, (size_t)extent * frame_in, ws->stream), "hipMemcpyAsync(H2D)") != 0
		   || cr_plan_launch(plan, w->d_in, extent * frame_in, w->d_out, 0, pf, n, ws->stream, out_s16) != 0;
		if (have_thread && !bad)
			bad = cr_check_hip(crhip_event_record(dl.slot[set].ready, ws->stream), "hipEventRecord") != 0;
		if (bad)
			break;

		if (have_thread)
		{
			pthread_mutex_lock(&dl.lock);
			dl.slot[set].host_dst = (unsigned char *)host_out + done * frame_out;
			dl.slot[set].dev_src = w->d_out;
			dl.slot[set].bytes = (size_t)n * frame_out;
			dl.slot[set].queued = 1;
			++dl.submitted;
			pthread_cond_broadcast(&dl.changed);
			pthread_mutex_unlock(&dl.lock);
		}
		else
		{
			bad = cr_check_hip(crhip_memcpy_d2h((unsigned char *)host_out + done * frame_out, w->d_out, (size_t)n * frame_out, ws->stream), "hipMemcpyAsync(D2H)") != 0
			   || cr_check_hip(crhip_stream_sync(ws->stream), "hipStreamSynchronize") != 0;
		}

		done += n;
		++batch;
	}

	if (have_thread)
	{
		int code;

		pthread_mutex_lock(&dl.lock);
		dl.quit = 1;
		pthread_cond_broadcast(&dl.changed);
		pthread_mutex_unlock(&dl.lock);
		pthread_join(thread, NULL);
		code = dl.hip_error;
		if (bad)
		{
			/* launches of a failed call may still be running: leave nothing in flight on the staging sets */
			crhip_stream_sync(ws->stream);
			crhip_stream_sync(g_workspace_more[0].stream);
		}
		if (code != 0 && !bad)
			bad = cr_check_hip(code, "download thread (hipMemcpyAsync D2H / hipStreamSynchronize)") != 0;
	}
	if (pipelined)
	{
		{
			int k;
			for (k = 0; k < 1 + CR_EXTRA_SETS; ++k)
				if (dl.slot[k].ready != NULL)
					crhip_event_destroy(dl.slot[k].ready);
		}
		pthread_cond_destroy(&dl.changed);
		pthread_mutex_destroy(&dl.lock);
	}

	cr_workspace_release(ws);
	return bad ? -1 : 0;
}

int cr_run_single_frame(const ClownResamplerAMD_Plan *plan, const int16_t *host_window, uint64_t window_frames,
                        uint64_t pos_frac, const int64_t *acc_in, int64_t *acc_out)
{
	const size_t in_bytes = (size_t)window_frames * plan->channels * sizeof(int16_t);
	const size_t acc_bytes = (size_t)plan->channels * sizeof(int64_t);
	cr_workspace *ws = cr_workspace_acquire(in_bytes, 2 * acc_bytes);
	crhip_generic_launch g;
	int bad;

	if (ws == NULL)
		return -1;

	memset(&g, 0, sizeof(g));
	g.d_in = ws->d_in;
	g.d_out = ws->d_out;
	g.d_table = plan->d_table;
	g.d_acc_in = (const int64_t *)(ws->d_out + acc_bytes);
	g.pos_int = 0;
	g.pos_frac = pos_frac;
	g.increment = plan->increment;
	g.n_out = 1;
	g.skr = plan->cfg.skr;
	g.radius_frames = plan->cfg.radius_frames;
	g.delta = plan->cfg.delta;
	g.step = plan->cfg.step;
	g.table_len = plan->table_len;
	g.channels = plan->channels;
	g.out64 = 1;

	bad = cr_check_hip(crhip_memcpy_h2d(ws->d_in, host_window, in_bytes, ws->stream), "hipMemcpyAsync(H2D)") != 0
	   || cr_check_hip(crhip_memcpy_h2d(ws->d_out + acc_bytes, acc_in, acc_bytes, ws->stream), "hipMemcpyAsync(H2D)") != 0
	   || cr_check_hip(crhip_launch_generic(&g, ws->stream), "k_generic launch") != 0
	   || cr_check_hip(crhip_memcpy_d2h(acc_out, ws->d_out, acc_bytes, ws->stream), "hipMemcpyAsync(D2H)") != 0
	   || cr_check_hip(crhip_stream_sync(ws->stream), "hipStreamSynchronize") != 0;

	cr_workspace_release(ws);
	return bad ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------------------- */
/* plan introspection and debug switches (public, radius-independent)                                                         */
/* ------------------------------------------------------------------------------------------------------- */

void ClownResamplerAMD_PlanGetInfo(const ClownResamplerAMD_Plan *plan, ClownResamplerAMD_PlanInfo *info)
{
	memset(info, 0, sizeof(*info));
	info->kernel = plan->use_poly ? (plan->vecs >= 200u ? 3u : plan->vecs >= 100u ? 2u : 1u) : 0u;
	info->variant = plan->variant;
	info->channels = plan->channels;
	info->norm_mode = plan->poly.norm_mode;
	info->slots = plan->poly.slots;
	info->first_slot = plan->poly.first_slot;
	info->rows = plan->poly.rows;
	info->row_stride = plan->poly.row_stride;
	info->row_mode = plan->poly.row_mode;
	info->threads = plan->threads;
	info->tile_frames = plan->tile_frames;
	info->lds_bytes = plan->lds_bytes;
	info->max_blocks = plan->max_blocks;
	info->specialised = plan->specialised;
}

const int32_t *ClownResamplerAMD_PlanRows(const ClownResamplerAMD_Plan *plan)
{
	return plan->poly.weights;
}

uint32_t ClownResamplerAMD_PlanRowOf(const ClownResamplerAMD_Plan *plan, uint32_t position_fractional)
{
	return cr_poly_row_of(&plan->poly, position_fractional & 0xFFFFu);
}

void ClownResamplerAMD_DebugSetVariant(int variant)
{
	g_variant = ((variant >= 0 && variant < crhip_poly_variants()) || (variant >= 1000 && variant < 1010)) ? variant : CR_DEFAULT_VARIANT;
}

void ClownResamplerAMD_DebugForceGenericKernel(int on)
{
	g_force_generic = on != 0;
}

/* ------------------------------------------------------------------------------------------------------- */
/* streaming side windows                                                                                  */
/* ------------------------------------------------------------------------------------------------------- */

cr_stream *cr_stream_create(void)
{
	cr_stream *stream = (cr_stream *)calloc(1, sizeof(*stream));

	if (stream == NULL)
		return NULL;

	pthread_mutex_lock(&g_lock);
	stream->id = ++g_stream_serial ^ 0x434C4F574E5253ull; /* never 0 */
	stream->next = g_streams;
	g_streams = stream;
	pthread_mutex_unlock(&g_lock);
	return stream;
}

cr_stream *cr_stream_lookup(uint64_t id)
{
	cr_stream *stream;

	pthread_mutex_lock(&g_lock);
	for (stream = g_streams; stream != NULL; stream = stream->next)
		if (stream->id == id)
			break;
	pthread_mutex_unlock(&g_lock);
	return stream;
}

int cr_stream_reserve(cr_stream *stream, size_t samples)
{
	int16_t *grown;

	if (stream->window_samples >= samples)
		return 0;

	grown = (int16_t *)realloc(stream->window, samples * sizeof(int16_t));
	if (grown == NULL)
		return -1;

	memset(grown + stream->window_samples, 0, (samples - stream->window_samples) * sizeof(int16_t));
	stream->window = grown;
	stream->window_samples = samples;
	return 0;
}

size_t cr_stream_max_frames(void)
{
	return g_stream_max_frames;
}

void ClownResamplerAMD_SetStreamingWindow(size_t frames)
{
	g_stream_max_frames = frames;
}

void ClownResamplerAMD_DebugSetStampBuffer(void *device_buffer)
{
	g_debug_stamps = (unsigned long long *)device_buffer;
}
