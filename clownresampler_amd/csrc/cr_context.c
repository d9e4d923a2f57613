/*
 * cr_context.c - process-wide GPU context: error reporting, device selection, plan cache, staging workspace.
 * Host C; reaches the GPU only through the crhip_* shim (crhip.h).  There is deliberately no CPU
 * implementation of the resampling arithmetic anywhere in this library: without a usable device every resample
 * entry point ends in cr_fail().
 */
#include "cr_context.h"

#include <pthread.h>
#include <signal.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

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
static __thread int t_deferred = 0;
static volatile sig_atomic_t g_abort_dumped = 0;   /* the flight recorder (below) has been written out once */

void cr_error_defer(int on)
{
	t_deferred = on;
}

int cr_error_take(char *message, size_t capacity)
{
	const int code = t_last_code;

	if (capacity != 0)
	{
		strncpy(message, code != 0 ? t_last_message : "", capacity - 1);
		message[capacity - 1] = '\0';
	}
	t_last_code = 0;
	t_last_message[0] = '\0';
	return code;
}

int cr_fail(int code, const char *format, ...)
{
	va_list ap;

	va_start(ap, format);
	vsnprintf(t_last_message, sizeof(t_last_message), format, ap);
	va_end(ap);
	t_last_code = code;
	++t_error_serial;

	/* a thread the LIBRARY created (the callback API's compute-ahead helper): the failure is recorded here and re-raised by the
	   client's own thread (cr_error_take / cr_fail there) - the client's handler never runs on a thread it did not create, and
	   ClownResamplerAMD_LastErrorCode on the calling thread tells the story */
	if (t_deferred)
		return code;

	if (g_handler != NULL)
	{
		g_handler(code, t_last_message, g_handler_user);
	}
	else
	{
		/* No handler installed.  The reference has no failing path (clownresampler.h:746-748: its two outcomes are "ran out of input" and
		   "the callback said stop"), and a drop-in must not end the host process over an error it can hand back: the failure is RECORDED
		   (ClownResamplerAMD_LastErrorCode / LastErrorMessage on this thread), said once on stderr - the first few of a process, so that a
		   stream that falls silent has an explanation - and the entry point returns the "callback said stop" outcome with nothing consumed
		   beyond what the consumer was given.  CLOWNRESAMPLER_AMD_ABORT_ON_ERROR=1 in the environment restores the hard stop (with the
		   flight recorder) for whoever prefers a core dump at the first failure. */
		static int said = 0;
		static int abort_on_error = -1;

		if (abort_on_error < 0)
		{
			const char *e = getenv("CLOWNRESAMPLER_AMD_ABORT_ON_ERROR");
			abort_on_error = (e != NULL && *e != '\0' && *e != '0') ? 1 : 0;
		}
		if (abort_on_error || __atomic_fetch_add(&said, 1, __ATOMIC_RELAXED) < 8)
		{
			fprintf(stderr, "clownresampler_amd: %s\n", t_last_message);
			fflush(stderr);
		}
		if (abort_on_error)
		{
			if (!g_abort_dumped)
			{
				g_abort_dumped = 1;
				ClownResamplerAMD_DebugDumpFlightRecorder(2);
			}
			abort();
		}
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
/* flight recorder                                                                                         */
/* ------------------------------------------------------------------------------------------------------- */

/* The last CR_FLIGHT_EVENTS things this process asked the device for through this library - kernel launches with every address
   range they may touch, device allocations and releases - in a ring that costs a launch ~100 bytes of stores.  What a process that
   dies of a GPU memory fault (reported by the runtime's own thread, a while after the launch returned) is otherwise unable to say:
   ClownResamplerAMD_DebugDumpFlightRecorder writes it out, cr_fail's default (abort) does so too, and a client - the GPU tests -
   may have it written when SIGABRT arrives (ClownResamplerAMD_DebugInstallAbortDump).  Nothing here allocates or locks. */
#define CR_FLIGHT_EVENTS 64u
enum { CR_FLIGHT_GENERIC = 0, CR_FLIGHT_POLY = 1, CR_FLIGHT_INT = 2, CR_FLIGHT_SEG = 3, CR_FLIGHT_TABLE = 4, CR_FLIGHT_MALLOC = 10, CR_FLIGHT_FREE = 11, CR_FLIGHT_HOST_ALLOC = 12, CR_FLIGHT_HOST_FREE = 13 };
typedef struct cr_flight_event
{
	unsigned long long serial;      /* 0: never written */
	uint32_t kind;
	int32_t result;                 /* what the HIP call returned */
	const void *d_in, *d_out, *aux, *tickets, *stream;   /* aux: rows image / table / segment table; for allocations d_out is the pointer */
	uint64_t in_bytes, out_bytes, aux_bytes, n_out, pos0;
	uint32_t increment, channels, slots, threads, blocks, lds_bytes, tile_frames, vecs, variant, flags;
} cr_flight_event;
static cr_flight_event g_flight[CR_FLIGHT_EVENTS];
static unsigned long long g_flight_serial = 0;

static cr_flight_event *flight_next(uint32_t kind)
{
	const unsigned long long serial = __atomic_add_fetch(&g_flight_serial, 1ull, __ATOMIC_RELAXED);
	cr_flight_event *e = &g_flight[serial % CR_FLIGHT_EVENTS];

	memset(e, 0, sizeof(*e));
	e->kind = kind;
	e->serial = serial;
	return e;
}

static void flight_memory(uint32_t kind, const void *pointer, size_t bytes, int result)
{
	cr_flight_event *e = flight_next(kind);

	e->d_out = pointer;
	e->out_bytes = bytes;
	e->result = result;
}

void ClownResamplerAMD_DebugDumpFlightRecorder(int fd)
{
	static const char *const names[] = {"k_generic", "k_poly-family", "k_int", "k_seg", "k_generic_segments"};
	const unsigned long long last = __atomic_load_n(&g_flight_serial, __ATOMIC_RELAXED);
	char line[640];
	unsigned k;
	int n;

	n = snprintf(line, sizeof(line), "clownresampler_amd: flight recorder, %llu events so far, the last %u oldest first (library %s)\n", last, CR_FLIGHT_EVENTS,
	             ClownResamplerAMD_BuildId());
	if (n > 0 && write(fd, line, (size_t)n) < 0)
		return;
	for (k = 0; k < CR_FLIGHT_EVENTS; ++k)
	{
		const cr_flight_event *e = &g_flight[(last + 1u + k) % CR_FLIGHT_EVENTS];

		if (e->serial == 0)
			continue;
		if (e->kind >= CR_FLIGHT_MALLOC)
			n = snprintf(line, sizeof(line), "  #%llu %s %p + %llu -> %d\n", e->serial,
			             e->kind == CR_FLIGHT_MALLOC ? "hipMalloc" : e->kind == CR_FLIGHT_FREE ? "hipFree" : e->kind == CR_FLIGHT_HOST_ALLOC ? "hipHostMalloc" : "hipHostFree",
			             e->d_out, (unsigned long long)e->out_bytes, (int)e->result);
		else
			n = snprintf(line, sizeof(line), "  #%llu %s -> %d: in %p + %llu, out %p + %llu, aux %p + %llu, tickets %p, stream %p, n_out %llu, pos0 0x%llx, increment 0x%x, "
			             "%u ch, %u slots, grid %u x %u, lds %u, tile %u, vecs %u, variant %u, flags 0x%x\n",
			             e->serial, names[e->kind <= CR_FLIGHT_TABLE ? e->kind : 0], (int)e->result, e->d_in, (unsigned long long)e->in_bytes, e->d_out, (unsigned long long)e->out_bytes,
			             e->aux, (unsigned long long)e->aux_bytes, e->tickets, e->stream, (unsigned long long)e->n_out, (unsigned long long)e->pos0, e->increment,
			             e->channels, e->slots, e->blocks, e->threads, e->lds_bytes, e->tile_frames, e->vecs, e->variant, e->flags);
		if (n > 0 && write(fd, line, (size_t)(n < (int)sizeof(line) ? n : (int)sizeof(line) - 1)) < 0)
			return;
	}
}

static struct sigaction g_abort_previous;
static int g_abort_installed = 0;

static void abort_dump(int sig, siginfo_t *info, void *context)
{
	if (!g_abort_dumped)
	{
		g_abort_dumped = 1;
		ClownResamplerAMD_DebugDumpFlightRecorder(2);
	}
	/* then whoever was there before (Python's faulthandler in the tests), else the default action */
	if ((g_abort_previous.sa_flags & SA_SIGINFO) && g_abort_previous.sa_sigaction != NULL)
		g_abort_previous.sa_sigaction(sig, info, context);
	else if (!(g_abort_previous.sa_flags & SA_SIGINFO) && g_abort_previous.sa_handler != SIG_DFL && g_abort_previous.sa_handler != SIG_IGN && g_abort_previous.sa_handler != NULL)
		g_abort_previous.sa_handler(sig);
	signal(SIGABRT, SIG_DFL);
	raise(SIGABRT);
}

int ClownResamplerAMD_DebugInstallAbortDump(void)
{
	struct sigaction sa;

	if (g_abort_installed)
		return 0;
	memset(&sa, 0, sizeof(sa));
	sa.sa_sigaction = abort_dump;
	sa.sa_flags = SA_SIGINFO | SA_NODEFER;
	sigemptyset(&sa.sa_mask);
	if (sigaction(SIGABRT, &sa, &g_abort_previous) != 0)
		return -1;
	g_abort_installed = 1;
	return 0;
}

/* ------------------------------------------------------------------------------------------------------- */
/* devices                                                                                                 */
/* ------------------------------------------------------------------------------------------------------- */

/* One context per HIP device, created on first use and kept until ClownResamplerAMD_Shutdown: its properties, its ticket
   blocks and its staging workspace.  Plans carry the ordinal of the device their rows live on.  Which device a call uses:
   the plan's, where the call is given a plan; otherwise the calling thread's (ClownResamplerAMD_SetThreadDevice) or, when the
   thread has none, the process default (ClownResamplerAMD_SetDevice).  Nothing here is torn down when the current device
   changes: a C client can drive all the GPUs of a node from one process, one thread per GPU or one thread for all. */
#define CR_MAX_DEVICES 64
#define CR_EXTRA_SETS 2 /* further staging sets of the pipelined host path (cr_run_host): three in all */
/* small host-buffer calls: one pinned, device-visible block (input at the front, output behind it) that the kernel reads and
   writes across PCIe itself - see cr_run_host */
#define CR_SMALL_CALL_BYTES (128u * 1024u)

/* Ticket blocks for the dynamic tile scheduling of k_poly / k_wave (crhip.h CRHIP_TICKET_WORDS counters each), zeroed once; a
   launch takes a block and leaves it zeroed.  Two launches that may run AT THE SAME TIME must not share a block.  Launches on
   one stream never do, so the blocks are handed out per stream: every stream seen gets a RING of CR_RING_SLOTS blocks (a ring
   rather than one block so that back-to-back launches of a stream, whose tails and heads overlap on the device, differ).
   A stream that has no ring takes an unused one, or takes over the ring of a stream that hipStreamQuery reports idle (its
   launches are through, their blocks are zeroed again; a stream the caller has destroyed counts as idle) - no device
   synchronise, ever - and when every ring belongs to a busy stream the table GROWS.
   Launches issued during a STREAM CAPTURE get blocks from a separate pool that is never recycled: the captured graph will
   reference its blocks for as long as it is replayed, side by side with whatever else is running then.  The pool is sized
   ahead of time (ClownResamplerAMD_ReserveCaptureLaunches; device memory cannot be allocated during a capture). */
#define CR_RING_SLOTS 64u
#define CR_RINGS_AT_START 8u
#define CR_CAPTURE_BLOCKS_DEFAULT 256u

typedef struct cr_ring
{
	void *stream;
	uint32_t *blocks;            /* CR_RING_SLOTS blocks of CRHIP_TICKET_WORDS, device memory */
	int used;
	unsigned next;
	unsigned pending;            /* launches that have drawn a block of this ring and are not enqueued yet: hipStreamQuery
	                                cannot see them, so a ring with any is never taken over */
	unsigned long long last_use;
} cr_ring;

typedef struct cr_device_ctx
{
	int ordinal;
	int ready;
	crhip_device_info info;
	/* tickets */
	pthread_mutex_t ring_lock;
	cr_ring *rings;
	unsigned ring_count;
	unsigned long long ring_clock;
	uint32_t **capture_chunks;   /* device allocations of the capture pool */
	unsigned capture_chunk_count;
	uint32_t *capture_at;        /* next unused block of the newest chunk */
	size_t capture_left;         /* blocks left in it */
	size_t capture_first_blocks; /* size of chunk 0 (ClownResamplerAMD_ReleaseCapturedLaunches rewinds to it) */
	/* staging (host-buffer entry points) */
	pthread_mutex_t workspace_lock;
	cr_workspace workspace;
	cr_workspace workspace_more[CR_EXTRA_SETS]; /* ([0].stream is the download stream) */
	unsigned char *small;
	/* the segment table of the one-launch variable-rate path (cr_segments_run): a pinned host copy and the device copy, reused
	   from call to call behind an event (under workspace_lock) */
	crhip_segment *seg_host, *seg_dev;
	size_t seg_capacity;
	void *seg_event;
	int seg_in_use;
} cr_device_ctx;

static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static cr_device_ctx *g_ctx[CR_MAX_DEVICES];
static int g_device = 0;                       /* process default */
static __thread int t_device = -1;             /* the calling thread's choice, -1: follow the process default */
static ClownResamplerAMD_Plan *g_plans = NULL;
static uint64_t g_plan_clock = 0;
static size_t g_plan_limit = 64;   /* unpinned plans kept; ClownResamplerAMD_SetPlanCacheLimit */
static int g_force_generic = 0;
static int g_no_int_kernel = 0;   /* ClownResamplerAMD_DebugDisableIntKernel */
static int g_no_dual_mono = 0;    /* ClownResamplerAMD_DebugDisableDualMono */
static int g_seg_mode = 0;        /* ClownResamplerAMD_DebugSegKernel: 0 the rule, 1 whenever the launch can take it, 2 never */
static unsigned long long *g_debug_stamps = NULL;
static int g_variant = -1; /* -1: CLOWNRESAMPLER_AMD_VARIANT from the environment, else the default */
/* launches enqueued so far, by kernel (numbered as ClownResamplerAMD_PlanInfo.kernel; 5 = k_int): what tests and bench.py
   assert "the kernel I mean is the one that ran" with (ClownResamplerAMD_DebugLaunchCount) */
/* ... and [7]: how many of those launches drew their tiles as TICKETS (k_poly with dynamic_tiles on, k_int with ticket groups) -
   the scheduler the long launches of cfg 2 / cfg 5 take and launches of fewer than eight tiles per workgroup do not */
#define CR_KERNEL_IDS 9
#define CR_COUNT_TICKETED 7
#define CR_COUNT_SEG 8            /* k_seg launches */
static unsigned long long g_launch_count[CR_KERNEL_IDS];

/* Where a plan WITHOUT a specialised instance runs the run-time-slot k_wave2 instead of the run-time-slot k_poly.  Measured on
   one box per pair, tools/channel_table.py with CLOWNRESAMPLER_AMD_RT_WAVE2_MIN_SLOTS=999 as the "off" leg
   (profiles/r02_rt_wave2.log): the 8-lobe build at 1.4:1 ... 3:1 (22-48 slots) gains +5 ... +52 % for 2-7 channels (two
   odd-channel cells lose 3-11 %); exact 3:1 with 3 lobes (18 slots, a handful of rows) +12 ... +42 %; but 3 lobes at 3.2:1
   (19 slots) -16 ... -25 %, at 4:1 -16 ... -36 %, 48 -> 11.025 and 44.1 -> 8 kHz mixed; mono and 8 channels lose at any length
   (too little arithmetic per frame for the per-tile work / too few waves beside the rows).  Hence: 2-7 channels, at most 3.25
   input frames per output frame, and 22 slots - or 18 where the rows are a few KB (exact ratios). */
/* k_wave2s (a lane per channel pair): from this many channels and slots on, plans without a specialised instance take it */
#define CR_WAVE2S_MIN_CHANNELS 99   /* never by default: measured slower than the run-time-slot k_poly (cr_kwave2s.hpp, profiles/r03_wave2s.log) */
#define CR_WAVE2S_MIN_SLOTS 12u
#define CR_RT_WAVE2_MIN_SLOTS 22   /* (20-21 slots within the increment limit would be 3 lobes at 3.2:1: the losing case) */
#define CR_RT_WAVE2_MIN_SLOTS_FEW_ROWS 18
#define CR_RT_WAVE2_FEW_ROWS_BYTES 8192u
#define CR_RT_WAVE2_MAX_INCREMENT ((13u << 16) / 4u)
/* k_up against its fallback kernel by launch length, in HALF wave-tiles per wave of k_up's grid (tools/brief_sweep.py,
   profiles/r02_brief_launches.log).  8 lobes, 8 -> 96 kHz: 10 s of it (half a tile per wave) 24.5 against k_wave2's 10.3 us, 3 tiles
   26.5 / 25.2, 4 tiles 28.1 / 31.4, and the same crossing at 8x and 10x; 3 lobes (k_wave the other kernel) 8x: 2 tiles 16.5 / 15.1,
   3 tiles 17.3 / 22.0, 16x: 1 tile 31.4 / 21.2, 2 tiles 31.9 / 35.5 */
#define CR_BRIEF_HALF_TILES_LONG_WINDOWS 7
#define CR_BRIEF_HALF_TILES 3
/* bytes of slack k_up2 keeps on either side of a wave's staged frames (= UP2_SLACK, cr_kup.hpp) */
/* k_poly's padded tiles (9-11, 13-15 channels without a specialised instance): up to 2:1 downsampling.  One box, 8 lobes: 44.1 -> 48
   kHz + 14-19 %, 48 -> 44.1 kHz + 3-17 %; at 44.1 -> 8 kHz, where 32-byte frames halve the tile, - 11 to + 13 % (profiles/r04_padded_tiles_ab.log) */
#define CR_PADDED_MAX_INCREMENT (2u << 16)
#define CR_UP2_SLACK 144u
#define CR_UP2_ENTRIES_BYTES 1280u
/* k_up2 as an instance's default kernel: see plan_geometry */
#define CR_UP_DEFAULT_MAX_INCREMENT (65536u / 8u)
#define CR_UP_DEFAULT_MIN_INCREMENT (65536u / 13u)
/* k_seg (cr_kseg.hpp): the ratios it is built for (any pure upsampling its instance has a ring for: crhip_seg_instance), the tile
   sizes, and how much of a launch its last, partial super-block may waste in idle lanes before the launch stays with k_up2 */
#define CR_SEG_MIN_INCREMENT 4096u
#define CR_SEG_MIN_TILE 64u
#define CR_SEG_MAX_TILE 128u
#define CR_SEG_MAX_WASTE 0.06
/* consecutive tiles an XCD takes per round (crhip_seg_launch.xcd_run; cr_kseg.hpp): neighbouring tiles read neighbouring bytes of the same input lines */
#define CR_SEG_XCD_RUN 16
/* a launch of a periodic ratio that starts mid-period is split (a few frames on the ordinary kernel, the rest on k_int) from this
   many output frames on: below, the second launch costs more than k_int saves */
#define CR_INT_SPLIT_MIN_FRAMES 8192u
/* (stereo 44.1 -> 48 kHz models 12 -> 4 and measured 0.5-1 % SLOWER rotated; exactly 8x models 28 -> 12, 16x 60 -> 12) */
#define CR_ROTATE_MIN_GAIN 12.0

/* Environment switches are read ONCE (tuning hooks; none of them changes results). */
static struct
{
	int loaded;
	int dynamic_tiles;          /* CLOWNRESAMPLER_AMD_DYNAMIC_TILES: -1 unset, else 0 / 1 */
	int w2_form;   /* diagnostic builds (-DCRA_WITH_W2_FORMS): k_wave2's timing-only forms */
	int no_special, debug, no_occupancy_clamp, tile_groups, no_host_pipeline, no_small_call_path;
	int no_replay_thread;       /* CLOWNRESAMPLER_AMD_NO_REPLAY_THREAD: the callback API never starts its compute-ahead helper thread */
	int no_dual_mono;           /* CLOWNRESAMPLER_AMD_NO_DUAL_MONO: long mono launches stay on the mono kernels (the A/B leg) */
	int no_padded_tiles;        /* CLOWNRESAMPLER_AMD_NO_PADDED_TILES: 9-11 / 13-15 channels compute from the tiles as the DMA leaves them (the A/B leg) */
	int host_direct;            /* CLOWNRESAMPLER_AMD_HOST_DIRECT: -1 unset (the rule), 0 never, 1 input only, 2 input and output - see cr_run_host */
	int wave2s_min_channels;    /* CLOWNRESAMPLER_AMD_WAVE2S_MIN_CHANNELS: frames from this many channels on take k_wave2s for long windows (99: never) */
	int lane_map;               /* CLOWNRESAMPLER_AMD_LANE_MAP: 0 / 1 forces k_wave2's lane order (unset: the conflict model picks) */
	int no_int_kernel;          /* CLOWNRESAMPLER_AMD_NO_INT_KERNEL: whole-number ratios take the plan's ordinary kernel (the A/B leg) */
	int rt_wave2_min_slots;     /* CLOWNRESAMPLER_AMD_RT_WAVE2_MIN_SLOTS: windows from this many slots on take the run-time-slot k_wave2 */
	double rotate_min_gain;     /* CLOWNRESAMPLER_AMD_ROTATE_MIN_GAIN: see plan_pick_rotation */
	int no_seg;                 /* CLOWNRESAMPLER_AMD_NO_SEG: long k_up2-shaped launches stay with k_up2 (the A/B leg) */
	int seg_form;               /* CLOWNRESAMPLER_AMD_SEG_FORM: diagnostic instance of k_seg (crhip_seg_launch.debug_form) */
	int seg_tile;               /* CLOWNRESAMPLER_AMD_SEG_TILE: k_seg's frames per lane and tile (a multiple of 16; unset: the rule) */
	int seg_xcd_run;            /* CLOWNRESAMPLER_AMD_SEG_XCD_RUN: k_seg's consecutive tiles per XCD and round (a multiple of 4; 0: one by one; unset: CR_SEG_XCD_RUN) */
	int brief_half_tiles;       /* CLOWNRESAMPLER_AMD_BRIEF_HALF_TILES: k_up launches of fewer half wave-tiles per wave take the plan's other kernel (0: none do; unset: per instance) */
} g_env;
static pthread_once_t g_env_once = PTHREAD_ONCE_INIT;

static void load_env(void)
{
	const char *e;

	e = getenv("CLOWNRESAMPLER_AMD_DYNAMIC_TILES");
	g_env.dynamic_tiles = (e != NULL && *e != '\0') ? (atoi(e) != 0) : -1;
	g_env.no_special = getenv("CLOWNRESAMPLER_AMD_NO_SPECIAL") != NULL;
	g_env.no_seg = getenv("CLOWNRESAMPLER_AMD_NO_SEG") != NULL;
	e = getenv("CLOWNRESAMPLER_AMD_W2_FORM");
	g_env.w2_form = (e != NULL && *e != '\0') ? atoi(e) : 0;
	e = getenv("CLOWNRESAMPLER_AMD_SEG_FORM");
	g_env.seg_form = (e != NULL && *e != '\0') ? atoi(e) : 0;
#ifndef CRA_WITH_W2_FORMS
	if (g_env.seg_form != 4)   /* (1 ... 3: timing-only forms, results wrong - a diagnostic build's; 4 = cycle stamps, results right) */
		g_env.seg_form = 0;
#endif
	e = getenv("CLOWNRESAMPLER_AMD_SEG_TILE");
	g_env.seg_tile = (e != NULL && *e != '\0') ? atoi(e) : 0;
	e = getenv("CLOWNRESAMPLER_AMD_SEG_XCD_RUN");
	g_env.seg_xcd_run = (e != NULL && *e != '\0' && atoi(e) >= 0 && atoi(e) % 4 == 0) ? atoi(e) : CR_SEG_XCD_RUN;
	g_env.debug = getenv("CLOWNRESAMPLER_AMD_DEBUG") != NULL;
	g_env.no_occupancy_clamp = getenv("CLOWNRESAMPLER_AMD_NO_OCCUPANCY_CLAMP") != NULL;
	e = getenv("CLOWNRESAMPLER_AMD_TILE_GROUPS");
	g_env.tile_groups = (e != NULL && atoi(e) > 0) ? atoi(e) : 0;
	g_env.no_host_pipeline = getenv("CLOWNRESAMPLER_AMD_NO_HOST_PIPELINE") != NULL;
	g_env.no_small_call_path = getenv("CLOWNRESAMPLER_AMD_NO_SMALL_CALL_PATH") != NULL;
	g_env.no_replay_thread = getenv("CLOWNRESAMPLER_AMD_NO_REPLAY_THREAD") != NULL;
	g_env.no_dual_mono = getenv("CLOWNRESAMPLER_AMD_NO_DUAL_MONO") != NULL;
	g_env.no_padded_tiles = getenv("CLOWNRESAMPLER_AMD_NO_PADDED_TILES") != NULL;
	e = getenv("CLOWNRESAMPLER_AMD_HOST_DIRECT");
	g_env.host_direct = (e != NULL && *e != '\0') ? atoi(e) : -1;
	g_env.no_int_kernel = getenv("CLOWNRESAMPLER_AMD_NO_INT_KERNEL") != NULL;
	e = getenv("CLOWNRESAMPLER_AMD_WAVE2S_MIN_CHANNELS");
	g_env.wave2s_min_channels = (e != NULL && atoi(e) > 0) ? atoi(e) : CR_WAVE2S_MIN_CHANNELS;
	e = getenv("CLOWNRESAMPLER_AMD_LANE_MAP");
	g_env.lane_map = (e != NULL && *e != '\0') ? (atoi(e) != 0) : -1;
	e = getenv("CLOWNRESAMPLER_AMD_RT_WAVE2_MIN_SLOTS");
	g_env.rt_wave2_min_slots = (e != NULL && atoi(e) > 0) ? atoi(e) : CR_RT_WAVE2_MIN_SLOTS;
	e = getenv("CLOWNRESAMPLER_AMD_ROTATE_MIN_GAIN");
	g_env.rotate_min_gain = (e != NULL && *e != '\0') ? atof(e) : -1.0;   /* (negative: unset - the rule's own thresholds, which differ for mono) */
	e = getenv("CLOWNRESAMPLER_AMD_BRIEF_HALF_TILES");
	g_env.brief_half_tiles = (e != NULL && *e != '\0' && atoi(e) >= 0) ? atoi(e) : -1;
	g_env.loaded = 1;
}

static void env_ready(void)
{
	pthread_once(&g_env_once, load_env);
}

static int current_device(void)
{
	return t_device >= 0 ? t_device : g_device;
}

int ClownResamplerAMD_DeviceCount(void)
{
	int count = 0;

	if (crhip_device_count(&count) != 0)
		return 0;

	return count;
}

/* Never reports through the error handler: this is the question a client asks BEFORE it commits to this library. */
int ClownResamplerAMD_IsUsable(void)
{
	crhip_device_info info;
	int count = 0;
	const int device = current_device();

	if (crhip_device_count(&count) != 0 || count <= 0 || device < 0 || device >= count)
		return 0;
	if (crhip_get_device_info(device, &info) != 0)
		return 0;
	/* the kernels are gfx950 code objects, nothing else */
	return strncmp(info.arch, "gfx950", 6) == 0;
}

/* ---- the device operations, recorded (flight recorder above) ---- */
static int dev_malloc(void **pointer, size_t bytes)
{
	const int code = crhip_malloc(pointer, bytes);

	flight_memory(CR_FLIGHT_MALLOC, code == 0 ? *pointer : NULL, bytes, code);
	return code;
}

static int dev_free(void *pointer)
{
	int code;

	if (pointer == NULL)
		return 0;
	code = crhip_free(pointer);
	flight_memory(CR_FLIGHT_FREE, pointer, 0, code);
	return code;
}

static int host_alloc(void **pointer, size_t bytes)
{
	const int code = crhip_host_alloc(pointer, bytes);

	flight_memory(CR_FLIGHT_HOST_ALLOC, code == 0 ? *pointer : NULL, bytes, code);
	return code;
}

static int host_free(void *pointer)
{
	const int code = crhip_host_free(pointer);

	flight_memory(CR_FLIGHT_HOST_FREE, pointer, 0, code);
	return code;
}

/* Copies between the device and PAGEABLE host memory (the caller's buffers of the host-pointer entry points, the library's own malloc'ed
   images) go out in pieces of at most 1 MiB.  The runtime's answer to a larger one is to page-lock the host range IN PLACE (hsa_amd_memory_lock,
   device address == host address) and let the copy engine work on the caller's heap (profiles/r06_copy_path.txt: "HSA Copy Using Pinned
   resource" above 1 MiB, "... Using Staging resource" up to it).  Every abort of a GPU test run that could be placed - three, rounds 5 and 6 -
   was a GPU page fault at an address inside the process's brk heap, reported while the main thread sat in such a copy (a 1.1-1.2 MB
   tensor.cpu() of the test), none of this library's launches having been given a host address (the flight recorder above).  What inside the
   runtime or the kernel driver goes wrong has not been established (HISTORY.md, round 6); what this library can do is keep ITS copies of a
   client's pageable memory off that path: pieces the runtime stages through its own pinned buffers.  CLOWNRESAMPLER_AMD_PAGEABLE_PIECE in the
   environment (bytes; 0: whole copies, the behaviour of rounds 1-5) is the A/B hook. */
#define CR_PAGEABLE_PIECE ((size_t)1 << 20)
static size_t pageable_piece(void)
{
	static size_t piece = (size_t)-1;

	if (piece == (size_t)-1)
	{
		const char *e = getenv("CLOWNRESAMPLER_AMD_PAGEABLE_PIECE");
		piece = (e != NULL && *e != '\0') ? (size_t)strtoull(e, NULL, 10) : CR_PAGEABLE_PIECE;
	}
	return piece;
}

static int copy_pageable_h2d(void *device_destination, const void *host_source, size_t bytes, void *stream)
{
	const size_t piece = pageable_piece();
	size_t at = 0;

	if (piece == 0 || bytes <= piece)
		return crhip_memcpy_h2d(device_destination, host_source, bytes, stream);
	while (at < bytes)
	{
		const size_t n = bytes - at < piece ? bytes - at : piece;
		const int code = crhip_memcpy_h2d((unsigned char *)device_destination + at, (const unsigned char *)host_source + at, n, stream);

		if (code != 0)
			return code;
		at += n;
	}
	return 0;
}

static int copy_pageable_d2h(void *host_destination, const void *device_source, size_t bytes, void *stream)
{
	const size_t piece = pageable_piece();
	size_t at = 0;

	if (piece == 0 || bytes <= piece)
		return crhip_memcpy_d2h(host_destination, device_source, bytes, stream);
	while (at < bytes)
	{
		const size_t n = bytes - at < piece ? bytes - at : piece;
		const int code = crhip_memcpy_d2h((unsigned char *)host_destination + at, (const unsigned char *)device_source + at, n, stream);

		if (code != 0)
			return code;
		at += n;
	}
	return 0;
}

static int launch_poly(const ClownResamplerAMD_Plan *plan, const crhip_poly_launch *l, void *stream)
{
	cr_flight_event *e = flight_next(CR_FLIGHT_POLY);

	e->d_in = l->d_in;
	e->in_bytes = l->in_valid_bytes;
	e->d_out = l->d_out;
	e->out_bytes = (l->dual ? (uint64_t)l->dual_out_frames + l->dual_valid_frames : l->n_out * l->channels) * (l->out_s16 ? 2u : 4u);
	e->aux = l->d_rows;
	e->aux_bytes = (uint64_t)l->plane_rows * l->row_stride * 4u;
	e->tickets = l->d_tickets;
	e->stream = stream;
	e->n_out = l->n_out;
	e->pos0 = l->pos0;
	e->increment = l->increment;
	e->channels = l->channels;
	e->slots = l->slots;
	e->threads = l->threads;
	e->blocks = l->blocks;
	e->lds_bytes = l->lds_bytes;
	e->tile_frames = l->tile_frames;
	e->vecs = l->vecs;
	e->variant = l->variant;
	e->flags = (l->specialised ? 1u : 0u) | (l->dual ? 2u : 0u) | (l->out_s16 ? 4u : 0u) | (l->dynamic_tiles ? 8u : 0u) | (l->padded ? 16u : 0u) | ((uint32_t)plan->device << 8);
	e->result = -9999;   /* (still inside the launch call) */
	return e->result = crhip_launch_poly(l, stream);
}

static int launch_int(const ClownResamplerAMD_Plan *plan, const crhip_int_launch *l, void *stream)
{
	cr_flight_event *e = flight_next(CR_FLIGHT_INT);

	e->d_in = l->d_in;
	e->in_bytes = l->in_valid_bytes;
	e->d_out = l->d_out;
	e->out_bytes = l->n_out * l->channels * (l->out_s16 ? 2u : 4u);
	e->tickets = l->d_tickets;
	e->stream = stream;
	e->n_out = l->n_out;
	e->pos0 = l->first_frame;
	e->increment = l->ratio;
	e->channels = l->channels;
	e->slots = l->slots;
	e->threads = plan->intk.shape.threads;
	e->blocks = l->blocks;
	e->variant = l->period;
	e->flags = (l->out_s16 ? 4u : 0u) | ((uint32_t)plan->device << 8);
	e->result = -9999;
	return e->result = crhip_launch_int(l, stream);
}

static int launch_seg(const ClownResamplerAMD_Plan *plan, const crhip_seg_launch *l, void *stream)
{
	cr_flight_event *e = flight_next(CR_FLIGHT_SEG);

	e->d_in = l->d_in;
	e->in_bytes = l->in_valid_bytes;
	e->d_out = l->d_out;
	e->out_bytes = l->n_out * 8u;
	e->aux = l->d_rows;
	e->aux_bytes = (uint64_t)plan->poly.rows * 64u;
	e->tickets = l->d_tickets;
	e->stream = stream;
	e->n_out = l->n_out;
	e->pos0 = l->pos0;
	e->increment = l->increment;
	e->channels = 2u;
	e->slots = l->slots;
	e->threads = plan->seg.threads;
	e->blocks = l->blocks;
	e->lds_bytes = plan->seg.lds_bytes;
	e->tile_frames = l->tile_frames;
	e->flags = (uint32_t)plan->device << 8;
	e->result = -9999;
	return e->result = crhip_launch_seg(l, stream);
}

static int launch_generic(const ClownResamplerAMD_Plan *plan, const crhip_generic_launch *l, uint64_t in_bytes, void *stream)
{
	cr_flight_event *e = flight_next(CR_FLIGHT_GENERIC);

	e->d_in = l->d_in;
	e->in_bytes = in_bytes;
	e->d_out = l->d_out;
	e->out_bytes = l->n_out * l->channels * (l->out64 == 1u ? 8u : l->out64 == 2u ? 2u : 4u);
	e->aux = l->d_table;
	e->aux_bytes = (uint64_t)l->table_len * 4u;
	e->stream = stream;
	e->n_out = l->n_out;
	e->pos0 = (l->pos_int << 16) + l->pos_frac;
	e->increment = (uint32_t)l->increment;
	e->channels = l->channels;
	e->slots = (uint32_t)l->radius_frames * 2u;
	e->threads = 256u;
	e->flags = (l->out64 << 2) | ((uint32_t)plan->device << 8);
	e->result = -9999;
	return e->result = crhip_launch_generic(l, stream);
}

static int launch_segments(const ClownResamplerAMD_Plan *plan, const crhip_segments_launch *l, void *stream)
{
	cr_flight_event *e = flight_next(CR_FLIGHT_TABLE);

	e->d_in = l->d_in;
	e->d_out = l->d_out;
	e->out_bytes = l->n_out * l->channels * (l->out_s16 ? 2u : 4u);
	e->aux = l->d_segments;
	e->aux_bytes = (uint64_t)l->n_segments * sizeof(crhip_segment);
	e->tickets = l->d_table;
	e->stream = stream;
	e->n_out = l->n_out;
	e->channels = l->channels;
	e->slots = l->n_segments;
	e->threads = 256u;
	e->flags = (l->out_s16 ? 4u : 0u) | ((uint32_t)plan->device << 8);
	e->result = -9999;
	return e->result = crhip_launch_segments(l, stream);
}

static int ring_alloc(cr_ring *ring)
{
	const size_t bytes = (size_t)CR_RING_SLOTS * CRHIP_TICKET_WORDS * sizeof(uint32_t);

	memset(ring, 0, sizeof(*ring));
	if (cr_check_hip(dev_malloc((void **)&ring->blocks, bytes), "hipMalloc(tickets)") != 0)
		return -1;
	if (cr_check_hip(crhip_memset(ring->blocks, 0, bytes, NULL), "hipMemset(tickets)") != 0
	 || cr_check_hip(crhip_stream_sync(NULL), "hipStreamSynchronize") != 0)
	{
		dev_free(ring->blocks);
		ring->blocks = NULL;
		return -1;
	}
	return 0;
}

/* rings [ctx->ring_count, want) come into being; ring_lock held (or the context not yet published) */
static int rings_grow(cr_device_ctx *ctx, unsigned want)
{
	cr_ring *grown = (cr_ring *)realloc(ctx->rings, (size_t)want * sizeof(cr_ring));

	if (grown == NULL)
		return cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "out of host memory");
	ctx->rings = grown;
	while (ctx->ring_count < want)
	{
		if (ring_alloc(&ctx->rings[ctx->ring_count]) != 0)
			return -1;
		++ctx->ring_count;
	}
	return 0;
}

/* one more chunk of never-recycled blocks for launches issued during stream captures; ring_lock held (or unpublished) */
static int capture_pool_grow(cr_device_ctx *ctx, size_t blocks)
{
	const size_t bytes = blocks * CRHIP_TICKET_WORDS * sizeof(uint32_t);
	uint32_t **list = (uint32_t **)realloc(ctx->capture_chunks, (ctx->capture_chunk_count + 1u) * sizeof(uint32_t *));
	uint32_t *chunk = NULL;

	if (list == NULL)
		return cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "out of host memory");
	ctx->capture_chunks = list;
	if (cr_check_hip(dev_malloc((void **)&chunk, bytes), "hipMalloc(capture tickets)") != 0)
		return -1;
	if (cr_check_hip(crhip_memset(chunk, 0, bytes, NULL), "hipMemset(capture tickets)") != 0
	 || cr_check_hip(crhip_stream_sync(NULL), "hipStreamSynchronize") != 0)
	{
		dev_free(chunk);
		return -1;
	}
	if (ctx->capture_chunk_count == 0)
		ctx->capture_first_blocks = blocks;
	ctx->capture_chunks[ctx->capture_chunk_count++] = chunk;
	ctx->capture_at = chunk;      /* (what was left of the previous chunk is abandoned: blocks are never handed out twice) */
	ctx->capture_left = blocks;
	return 0;
}

/* g_lock held.  Selects the device for the calling thread (hipSetDevice is per thread) and makes sure its context exists. */
static cr_device_ctx *ensure_ctx_locked(int ordinal)
{
	cr_device_ctx *ctx;
	int count = 0;
	int e;

	if (ordinal >= 0 && ordinal < CR_MAX_DEVICES && g_ctx[ordinal] != NULL && g_ctx[ordinal]->ready)
		return cr_check_hip(crhip_set_device(ordinal), "hipSetDevice") == 0 ? g_ctx[ordinal] : NULL;

	env_ready();
	e = crhip_device_count(&count);

	if (e != 0 || count <= 0)
	{
		cr_fail(CLOWNRESAMPLER_AMD_ERROR_NO_DEVICE,
		        "no usable HIP device (%s); this library has no CPU fallback for the resampling path",
		        e != 0 ? crhip_error_string(e) : "device count is 0");
		return NULL;
	}

	if (ordinal < 0 || ordinal >= count || ordinal >= CR_MAX_DEVICES)
	{
		cr_fail(CLOWNRESAMPLER_AMD_ERROR_NO_DEVICE, "device %d selected but only %d present", ordinal, count);
		return NULL;
	}

	if (cr_check_hip(crhip_set_device(ordinal), "hipSetDevice") != 0)
		return NULL;

	ctx = g_ctx[ordinal];
	if (ctx == NULL)
	{
		ctx = (cr_device_ctx *)calloc(1, sizeof(*ctx));
		if (ctx == NULL)
		{
			cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "out of host memory");
			return NULL;
		}
		ctx->ordinal = ordinal;
		pthread_mutex_init(&ctx->ring_lock, NULL);
		pthread_mutex_init(&ctx->workspace_lock, NULL);
		__atomic_store_n(&g_ctx[ordinal], ctx, __ATOMIC_RELEASE);
	}

	if (cr_check_hip(crhip_get_device_info(ordinal, &ctx->info), "hipGetDeviceProperties") != 0)
		return NULL;

	if (ctx->ring_count < CR_RINGS_AT_START && rings_grow(ctx, CR_RINGS_AT_START) != 0)
		return NULL;
	if (ctx->capture_chunk_count == 0 && capture_pool_grow(ctx, CR_CAPTURE_BLOCKS_DEFAULT) != 0)
		return NULL;

	/* published last, with release semantics: the lock-free fast path of ensure_ctx reads it with an acquire load and must then
	   see the rings, the capture pool and `info` complete */
	__atomic_store_n(&ctx->ready, 1, __ATOMIC_RELEASE);
	return ctx;
}

static cr_device_ctx *ensure_ctx(int ordinal)
{
	cr_device_ctx *ctx;

	/* fast path without the process lock: a context, once ready, stays until Shutdown (which needs a quiescent library) */
	if (ordinal >= 0 && ordinal < CR_MAX_DEVICES)
	{
		ctx = __atomic_load_n(&g_ctx[ordinal], __ATOMIC_ACQUIRE);
		if (ctx != NULL && __atomic_load_n(&ctx->ready, __ATOMIC_ACQUIRE))
			return cr_check_hip(crhip_set_device(ordinal), "hipSetDevice") == 0 ? ctx : NULL;
	}

	pthread_mutex_lock(&g_lock);
	ctx = ensure_ctx_locked(ordinal);
	pthread_mutex_unlock(&g_lock);
	return ctx;
}

int cr_ensure_device(void)
{
	return ensure_ctx(current_device()) != NULL ? 0 : -1;
}

int cr_ensure_device_of(const ClownResamplerAMD_Plan *plan)
{
	return ensure_ctx(plan->device) != NULL ? 0 : -1;
}

int cr_current_device(void)
{
	return current_device();
}

/* *ring_out: the ring the block came from (its `pending` has been raised: ticket_block_enqueued() once the launch is in the
   stream's queue, whether or not it succeeded), or -1 for a block of the capture pool */
static uint32_t *ticket_block_for(cr_device_ctx *ctx, void *stream, int *ring_out)
{
	unsigned r, pick;
	uint32_t *block = NULL;
	int capturing = 0;

	*ring_out = -1;

	if (stream != NULL && crhip_stream_is_capturing(stream, &capturing) != 0)
		capturing = 0;

	pthread_mutex_lock(&ctx->ring_lock);

	if (capturing)
	{
		if (ctx->capture_left != 0)
		{
			block = ctx->capture_at;
			ctx->capture_at += CRHIP_TICKET_WORDS;
			--ctx->capture_left;
		}
		pthread_mutex_unlock(&ctx->ring_lock);
		if (block == NULL)
			cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "more launches captured into graphs than ticket blocks were set aside for (device memory cannot be "
			        "allocated during a capture): call ClownResamplerAMD_ReserveCaptureLaunches(n) before capturing");
		return block;
	}

	pick = ctx->ring_count;
	for (r = 0; r < ctx->ring_count; ++r)
	{
		if (ctx->rings[r].used && ctx->rings[r].stream == stream)
		{
			pick = r;
			break;
		}
		if (!ctx->rings[r].used && pick == ctx->ring_count)
			pick = r;
	}

	if (pick == ctx->ring_count)
	{
		/* every ring belongs to another stream: take over one whose stream has nothing left in flight, least recently used
		   first (a stream handle the caller has destroyed makes the query fail: that ring is free as well) */
		unsigned long long oldest = ~0ull;

		for (r = 0; r < ctx->ring_count; ++r)
		{
			int other_capturing = 0;

			/* not a ring some thread has drawn from without having enqueued its launch yet (the stream still reads idle), and
			   not one whose stream is recording a graph (hipStreamQuery is not allowed on a capturing stream and would
			   invalidate the capture) */
			if (ctx->rings[r].pending != 0 || ctx->rings[r].last_use >= oldest)
				continue;
			if (ctx->rings[r].stream != NULL && crhip_stream_is_capturing(ctx->rings[r].stream, &other_capturing) == 0 && other_capturing)
				continue;
			if (crhip_stream_busy(ctx->rings[r].stream) == 0)
			{
				oldest = ctx->rings[r].last_use;
				pick = r;
			}
		}

		if (pick == ctx->ring_count)
		{
			/* all busy: more rings (rare: more than ring_count streams with launches in flight at once) */
			if (rings_grow(ctx, ctx->ring_count + 4u) != 0)
			{
				pthread_mutex_unlock(&ctx->ring_lock);
				return NULL;
			}
		}
		ctx->rings[pick].used = 0;
	}

	if (!ctx->rings[pick].used)
	{
		/* (`next` goes on from where the previous owner left it: the new owner's first launch does not land on the slot the
		   old owner's first launches used) */
		ctx->rings[pick].used = 1;
		ctx->rings[pick].stream = stream;
	}
	ctx->rings[pick].last_use = ++ctx->ring_clock;
	++ctx->rings[pick].pending;
	*ring_out = (int)pick;
	block = ctx->rings[pick].blocks + (size_t)CRHIP_TICKET_WORDS * (ctx->rings[pick].next++ % CR_RING_SLOTS);
	pthread_mutex_unlock(&ctx->ring_lock);
	return block;
}

static void ticket_block_enqueued(cr_device_ctx *ctx, int ring)
{
	if (ring < 0)
		return;
	pthread_mutex_lock(&ctx->ring_lock);
	if ((unsigned)ring < ctx->ring_count && ctx->rings[ring].pending != 0)
		--ctx->rings[ring].pending;
	pthread_mutex_unlock(&ctx->ring_lock);
}

int ClownResamplerAMD_ReserveCaptureLaunches(size_t launches)
{
	cr_device_ctx *ctx = ensure_ctx(current_device());
	int r = 0;

	if (ctx == NULL)
		return -1;
	pthread_mutex_lock(&ctx->ring_lock);
	if (ctx->capture_left < launches)
		r = capture_pool_grow(ctx, launches);
	pthread_mutex_unlock(&ctx->ring_lock);
	return r != 0 ? -1 : 0;
}

/* The other end of the capture pool: a client that re-captures graphs over time says when the old ones are gone.
   Every block handed out to a captured launch on the current device becomes available again: the pool rewinds to its first
   chunk (zeroed again - a graph destroyed mid-flight could have left counters behind) and the later chunks are freed. */
int ClownResamplerAMD_ReleaseCapturedLaunches(void)
{
	cr_device_ctx *ctx = ensure_ctx(current_device());
	int r = 0;

	if (ctx == NULL)
		return -1;
	pthread_mutex_lock(&ctx->ring_lock);
	if (ctx->capture_chunk_count != 0)
	{
		const size_t bytes = ctx->capture_first_blocks * CRHIP_TICKET_WORDS * sizeof(uint32_t);

		while (ctx->capture_chunk_count > 1u)
			dev_free(ctx->capture_chunks[--ctx->capture_chunk_count]);   /* (hipFree waits for work that still uses it) */
		if (cr_check_hip(crhip_memset(ctx->capture_chunks[0], 0, bytes, NULL), "hipMemset(capture tickets)") != 0
		 || cr_check_hip(crhip_stream_sync(NULL), "hipStreamSynchronize") != 0)
			r = -1;
		ctx->capture_at = ctx->capture_chunks[0];
		ctx->capture_left = ctx->capture_first_blocks;
	}
	pthread_mutex_unlock(&ctx->ring_lock);
	return r;
}

const crhip_device_info *cr_device_info(void)
{
	const cr_device_ctx *ctx = ensure_ctx(current_device());
	static const crhip_device_info none;

	return ctx != NULL ? &ctx->info : &none;
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
		dev_free(store->d_table);
		dev_free(store->d_rows);
		dev_free(store->d_rows_seg);
	}
	cr_poly_free(&store->poly);
	free(store);
}

/* g_lock held.  A plan that has left the cache: its share of the rows, its private dual-mono partner, itself. */
static void plan_free(ClownResamplerAMD_Plan *plan, int device_usable)
{
	if (plan->dual.partner != NULL)
	{
		store_release(plan->dual.partner->store, device_usable);
		free(plan->dual.partner);
	}
	store_release(plan->store, device_usable);
	free(plan);
}

/* g_lock held.  Frees what one device holds; fails (returns -1, nothing freed) while a call still holds one of its plans. */
static int release_device_locked(cr_device_ctx *ctx)
{
	ClownResamplerAMD_Plan **link;
	unsigned k;

	for (link = &g_plans; *link != NULL; link = &(*link)->next)
		if ((*link)->device == ctx->ordinal && (*link)->users != 0)
			return -1;

	/* no host-buffer call may be inside the staging workspace, no launch may be drawing a ticket block */
	pthread_mutex_lock(&ctx->workspace_lock);
	pthread_mutex_lock(&ctx->ring_lock);

	if (ctx->ready)
		crhip_set_device(ctx->ordinal);

	link = &g_plans;
	while (*link != NULL)
	{
		ClownResamplerAMD_Plan *plan = *link;

		if (plan->device != ctx->ordinal)
		{
			link = &plan->next;
			continue;
		}
		*link = plan->next;
		plan_free(plan, ctx->ready);
	}

	if (ctx->ready)
	{
		for (k = 0; k < ctx->ring_count; ++k)
			dev_free(ctx->rings[k].blocks);
		for (k = 0; k < ctx->capture_chunk_count; ++k)
			dev_free(ctx->capture_chunks[k]);
		dev_free(ctx->workspace.d_in);
		dev_free(ctx->workspace.d_out);
		if (ctx->workspace.stream != NULL)
			crhip_stream_destroy(ctx->workspace.stream);
		if (ctx->small != NULL)
			host_free(ctx->small);
		if (ctx->seg_host != NULL)
			host_free(ctx->seg_host);
		dev_free(ctx->seg_dev);
		if (ctx->seg_event != NULL)
			crhip_event_destroy(ctx->seg_event);
		for (k = 0; k < CR_EXTRA_SETS; ++k)
		{
			dev_free(ctx->workspace_more[k].d_in);
			dev_free(ctx->workspace_more[k].d_out);
			if (ctx->workspace_more[k].stream != NULL)
				crhip_stream_destroy(ctx->workspace_more[k].stream);
		}
	}
	free(ctx->rings);
	free(ctx->capture_chunks);
	ctx->rings = NULL;
	ctx->ring_count = 0;
	ctx->capture_chunks = NULL;
	ctx->capture_chunk_count = 0;
	ctx->capture_at = NULL;
	ctx->capture_left = 0;
	ctx->small = NULL;
	ctx->seg_host = NULL;
	ctx->seg_dev = NULL;
	ctx->seg_capacity = 0;
	ctx->seg_event = NULL;
	ctx->seg_in_use = 0;
	memset(&ctx->workspace, 0, sizeof(ctx->workspace));
	memset(ctx->workspace_more, 0, sizeof(ctx->workspace_more));
	__atomic_store_n(&ctx->ready, 0, __ATOMIC_RELEASE);

	pthread_mutex_unlock(&ctx->ring_lock);
	pthread_mutex_unlock(&ctx->workspace_lock);
	return 0;
}

int ClownResamplerAMD_SetDevice(int ordinal)
{
	cr_device_ctx *ctx;

	pthread_mutex_lock(&g_lock);
	ctx = ensure_ctx_locked(ordinal);
	if (ctx != NULL)
		g_device = ordinal;
	pthread_mutex_unlock(&g_lock);
	return ctx != NULL ? 0 : -1;
}

int ClownResamplerAMD_GetDevice(void)
{
	return current_device();
}

int ClownResamplerAMD_SetThreadDevice(int ordinal)
{
	if (ordinal < 0)
	{
		t_device = -1;
		return 0;
	}
	if (ensure_ctx(ordinal) == NULL)
		return -1;
	t_device = ordinal;
	return 0;
}

/* Needs a QUIESCENT library: no resample call in progress on any thread (a plan still held by a call makes it fail loudly
   instead of freeing memory under that call). */
void ClownResamplerAMD_Shutdown(void)
{
	int d, busy = 0;

	cr_multi_shutdown();
	pthread_mutex_lock(&g_lock);
	for (d = 0; d < CR_MAX_DEVICES; ++d)
		if (g_ctx[d] != NULL && release_device_locked(g_ctx[d]) != 0)
			busy = 1;
	if (!busy)
	{
		while (g_streams != NULL)
		{
			cr_stream *next = g_streams->next;
			free(g_streams->window);
			free(g_streams->pull_ends);
			free(g_streams);
			g_streams = next;
		}
	}
	pthread_mutex_unlock(&g_lock);
	if (busy)
		cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "ClownResamplerAMD_Shutdown while a resample call is in progress on another thread: nothing of that device was released");
}

void *ClownResamplerAMD_DeviceAlloc(size_t bytes)
{
	void *p = NULL;

	if (cr_ensure_device() != 0)
		return NULL;
	if (cr_check_hip(dev_malloc(&p, bytes != 0 ? bytes : 16), "hipMalloc") != 0)
		return NULL;
	return p;
}

void ClownResamplerAMD_DeviceFree(void *device_pointer)
{
	if (device_pointer != NULL && cr_ensure_device() == 0)
		cr_check_hip(dev_free(device_pointer), "hipFree");
}

int ClownResamplerAMD_CopyToDevice(void *device_destination, const void *host_source, size_t bytes)
{
	if (cr_ensure_device() != 0)
		return -1;
	if (cr_check_hip(copy_pageable_h2d(device_destination, host_source, bytes, NULL), "hipMemcpyAsync(H2D)") != 0)
		return -1;
	return cr_check_hip(crhip_stream_sync(NULL), "hipStreamSynchronize") != 0 ? -1 : 0;
}

int ClownResamplerAMD_CopyFromDevice(void *host_destination, const void *device_source, size_t bytes)
{
	if (cr_ensure_device() != 0)
		return -1;
	if (cr_check_hip(copy_pageable_d2h(host_destination, device_source, bytes, NULL), "hipMemcpyAsync(D2H)") != 0)
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

/* Variants from 1000 up are DIAGNOSTIC instances.  1006 ... 1008 compute what the kernel they stand for computes (they add clock stamps); the
   others are timing-only forms whose RESULTS ARE WRONG by design - those a shipped library never takes, whatever the environment says: only
   a diagnostic build (make CRA_CFLAGS=-DCRA_WITH_W2_FORMS) accepts them. */
static int diagnostic_variant(int variant)
{
#ifdef CRA_WITH_W2_FORMS
	return variant >= 1000 && variant <= 1013;
#else
	return variant >= 1006 && variant <= 1008;
#endif
}

static uint32_t current_variant(void)
{
	if (g_variant < 0)
	{
		const char *e = getenv("CLOWNRESAMPLER_AMD_VARIANT");
		g_variant = (e != NULL && *e != '\0') ? atoi(e) : CR_DEFAULT_VARIANT;
		if (g_variant < 0 || (g_variant >= crhip_poly_variants() && !diagnostic_variant(g_variant)))
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
	                    && !g_env.no_special;
	return special ? plan->poly.row_stride : 4u * ((plan->poly.slots + 3u) / 4u + 1u);
}

/* Launch geometry of k_poly for this plan on the current device. */
static void plan_geometry(ClownResamplerAMD_Plan *plan)
{
	uint32_t frame_bytes = plan->channels * 2u;   /* (k_poly's padded tiles - 9-11, 13-15 channels without a specialised instance: 32 in the LDS tiles, below) */
	/* int32 per row of the device image: see cr_poly_device_image */
	const uint32_t image_stride = plan_image_stride(plan);
	const uint32_t rows_bytes = cr_poly_plane_rows(&plan->poly) * image_stride * 4u;
	uint32_t tile_bytes, cap_frames, per_cu;
	uint64_t tile;

	uint32_t frames_multiple = 0;
	const crhip_device_info *di = &g_ctx[plan->device]->info;
	/* frames of one tap window as a tile has to hold it: the slots plus the largest shift of a phase's window (shifted rows) */
	const uint32_t window_slots = plan->poly.slots + plan->poly.window_extra;

	plan->specialised = (uint32_t)crhip_poly_has_instance(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode);
	if (g_env.no_special) /* tuning hook (CLOWNRESAMPLER_AMD_NO_SPECIAL): time the run-time-slot instance instead */
		plan->specialised = 0;
	/* (variant 31 - the run-time-slot k_wave2, a testing hook as an explicit choice - means nothing to a specialised instance) */
	if (plan->variant == CRHIP_VARIANT_RT_WAVE2 && (plan->specialised || !crhip_poly_runtime_wave2(plan->channels, plan->poly.row_mode)))
		plan->variant = 0xFFFFu;
	if (plan->variant == CRHIP_VARIANT_RT_WAVE2S && (plan->specialised || !crhip_poly_runtime_wave2s(plan->channels)))
		plan->variant = 0xFFFFu;   /* (likewise variant 32, k_wave2s) */
	if (plan->specialised && plan->variant == 0xFFFFu && crhip_poly_has_up(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode))
	{
		/* Instances with the input-stationary kernel (k_up2: stereo, pure upsampling) - which ratios it is the default for
		   (tools/up_ratio_sweep.py, 40 M output frames, profiles/r02_up_ratios.log): */
		if (crhip_poly_default_is_up(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode))
		{
			/* 8 lobes: 8x..13x.  Below, a lane's handful of frames per input position does not pay for the window it unpacks
			   (2x: 196 against k_wave2's 124 us, 4x: 137 / 131, 6x: 119 / 118, 7.5x: 123 / 118, 8x: 116 / 126, 10x: 112 / 118,
			   12x: 109 / 121, 13x: 116 / 117); above, the wave-tile outgrows its LDS staging (15x: 125 / 117, 16x: 144 / 128) */
			if (plan->increment > CR_UP_DEFAULT_MAX_INCREMENT || plan->increment < CR_UP_DEFAULT_MIN_INCREMENT)
				plan->variant = crhip_poly_up_fallback_variant(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode);
		}
		/* (the 3-lobe stereo instance keeps k_up2 behind variant 27 only: its default is k_poly's chain, with its rows rotated
		   in LDS at the ratios where k_up2 used to win - exactly 8x: 79 against 80-90 us, 16x: 79 / 92) */
	}
	crhip_poly_geometry(plan->channels, plan->specialised ? plan->poly.slots : 0xFFFFu, plan->poly.row_mode, plan->poly.norm_mode, plan->variant, &plan->threads, &plan->vecs, &frames_multiple);
	if (plan->variant == 28u || plan->variant == 29u || (plan->variant == 0xFFFFu && crhip_poly_default_is_mad(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode)))
	{
		/* k_poly with the 64-bit multiply-add chain: compiled for fixed weight signs per slot, like k_up */
		uint32_t negmask = 0, pos_bits = 0, neg_bits = 0;
		const int any_sign = crhip_poly_mad_any_sign(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode);
		int ok = any_sign || crhip_poly_up_negmask(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode, &negmask);

		if (ok && !any_sign)
		{
			cr_poly_slot_signs(&plan->poly, &pos_bits, &neg_bits);
			ok = (neg_bits & ~negmask) == 0 && (pos_bits & negmask) == 0
			  && (cr_poly_slots_reaching(&plan->poly, 65536) & ~crhip_poly_mad_safemask(plan->poly.slots)) == 0;
		}

		if (!ok)
		{
			plan->variant = crhip_poly_fallback_variant();
			crhip_poly_geometry(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode, plan->variant, &plan->threads, &plan->vecs, &frames_multiple);
		}
	}

	if (!plan->specialised && crhip_poly_runtime_wave2s(plan->channels)
	 && (plan->variant == CRHIP_VARIANT_RT_WAVE2S
	     || (plan->variant == 0xFFFFu && plan->channels >= (uint32_t)g_env.wave2s_min_channels && plan->poly.slots >= CR_WAVE2S_MIN_SLOTS)))
	{
		/* k_wave2s: a lane per channel PAIR (cr_kwave2s.hpp) - wide frames with long windows, which had the run-time-slot k_poly
		   (two lanes per frame, ~5.6 VALU per tap and channel).  A wave-instruction covers 64 / ceil(channels / 2) frames; the
		   wave-tile is the largest multiple of that, up to 64 frames, whose window still leaves room for 8 waves beside the rows
		   (fewer frames per tile for heavy downsampling: 16 channels at 44.1 -> 8 kHz are 176 bytes of window per output frame). */
		const uint32_t per_instruction = 64u / ((plan->channels + 1u) / 2u);
		const uint32_t lds = (uint32_t)di->max_lds_per_block < 160u * 1024u ? (uint32_t)di->max_lds_per_block : 160u * 1024u;
		const uint32_t room = lds > rows_bytes + 16u ? lds - rows_bytes - 16u : 0u;
		uint32_t wt, best_wt = 0, best_waves = 0, best_pieces = 0;

		for (wt = (64u / per_instruction) * per_instruction; wt >= per_instruction; wt -= per_instruction)
		{
			const uint64_t last_rel = (65535u + (uint64_t)(wt - 1u) * plan->increment) >> 16;
			const uint64_t window = 12u + (last_rel + window_slots) * frame_bytes;
			const uint32_t pieces = (uint32_t)((window + 1023u) / 1024u);
			uint32_t waves = room / (4u * 1024u * pieces);

			if (waves > 16u)
				waves = 16u;
			if (pieces <= 8u && last_rel + window_slots < 65536u && (waves > best_waves && best_waves < 8u))
			{
				best_wt = wt;
				best_waves = waves;
				best_pieces = pieces;
			}
		}
		if (best_waves >= 4u && (uint64_t)best_wt * plan->increment < (1ull << 32) - 65536u)
		{
			plan->variant = CRHIP_VARIANT_RT_WAVE2S;
			plan->threads = best_waves * 64u;
			plan->vecs = 150u + best_pieces;
			plan->wave_tile = best_wt;
			plan->tile_frames = 4u * best_wt;
			plan->lds_bytes = rows_bytes + best_waves * 4u * 1024u * best_pieces + 16u;
			plan->max_blocks = (uint32_t)(di->compute_units > 0 ? di->compute_units : 256);
			return;
		}
	}
	if (plan->variant == CRHIP_VARIANT_RT_WAVE2S)
	{
		/* asked for, but the window does not fit: the ordinary choice */
		plan->variant = 0xFFFFu;
		crhip_poly_geometry(plan->channels, 0xFFFFu, plan->poly.row_mode, plan->poly.norm_mode, plan->variant, &plan->threads, &plan->vecs, &frames_multiple);
	}

	if (!plan->specialised && crhip_poly_runtime_wave2(plan->channels, plan->poly.row_mode)
	 && (plan->variant == CRHIP_VARIANT_RT_WAVE2
	     || (plan->variant == 0xFFFFu && plan->channels >= 2u && plan->channels <= 7u && plan->increment <= CR_RT_WAVE2_MAX_INCREMENT
	         && (plan->poly.slots >= (uint32_t)g_env.rt_wave2_min_slots
	             || (rows_bytes <= CR_RT_WAVE2_FEW_ROWS_BYTES && plan->poly.slots >= CR_RT_WAVE2_MIN_SLOTS_FEW_ROWS
	                 && g_env.rt_wave2_min_slots == CR_RT_WAVE2_MIN_SLOTS)))))
	{
		/* The run-time-slot k_wave2 (2 VALU per tap and channel instead of the ~5.6 of the run-time-slot k_poly loop), for
		   long windows: one frame per lane and wave-tile, as many 1 KiB window pieces as that takes, as many waves as then
		   fit beside the rows - at least 8 (two per SIMD), or the shape stays with k_poly. */
		const uint64_t last_rel = (65535u + 63ull * plan->increment) >> 16;
		const uint64_t window = 12u + (last_rel + window_slots) * frame_bytes;
		const uint32_t pieces = (uint32_t)((window + 1023u) / 1024u);
		const uint32_t lds = (uint32_t)di->max_lds_per_block < 160u * 1024u ? (uint32_t)di->max_lds_per_block : 160u * 1024u;
		const uint32_t room = lds > rows_bytes + 16u ? lds - rows_bytes - 16u : 0u;
		uint32_t waves = pieces > 0 ? room / (4u * 1024u * pieces) : 0u;

		if (waves > 16u)
			waves = 16u;
		if (pieces <= 8u && waves >= 8u && 64ull * plan->increment < (1ull << 32) - 65536u)
		{
			plan->variant = CRHIP_VARIANT_RT_WAVE2;
			plan->threads = waves * 64u;
			plan->vecs = 150u + pieces;
			plan->tile_frames = 256u;
			plan->lds_bytes = rows_bytes + waves * 4u * 1024u * pieces + 16u;
			plan->max_blocks = (uint32_t)(di->compute_units > 0 ? di->compute_units : 256);
			return;
		}
	}
	if (plan->variant == CRHIP_VARIANT_RT_WAVE2)
	{
		/* asked for, but the window does not fit: the run-time-slot k_poly */
		plan->variant = 0xFFFFu;
		crhip_poly_geometry(plan->channels, 0xFFFFu, plan->poly.row_mode, plan->poly.norm_mode, plan->variant, &plan->threads, &plan->vecs, &frames_multiple);
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
			/* k_up2 stages every weight outside the two centre slots as |weight| << 15 (the mov-armed chain, cr_kup.hpp) */
			if (ok && (cr_poly_slots_reaching(&plan->poly, 65536) & ~crhip_poly_mad_safemask(plan->poly.slots)) != 0)
				ok = 0;
		}

		if (wave_tile > frames_multiple)
			wave_tile = frames_multiple;
		{
			/* ... and no larger than the staging space one workgroup per CU leaves each wave (k_up2: slack on either side of a wave's
			   staged frames for the surplus frames of its unpredicated loop, UP2_SLACK in cr_kup.hpp) */
			const uint32_t waves = plan->threads / 64u;
			const uint32_t fixed = rows_bytes + 16u + waves * (2u * 1024u + 2u * CR_UP2_SLACK);
			const uint32_t lds = (uint32_t)di->max_lds_per_block < 160u * 1024u ? (uint32_t)di->max_lds_per_block : 160u * 1024u;
			const uint64_t room = lds > fixed ? ((lds - fixed) / waves & ~15u) / unit : 0;

			if (wave_tile > room)
				wave_tile = room;
			if (wave_tile < 64u)
				ok = 0;
		}

		if (ok)
		{
			uint32_t stage_bytes = ((uint32_t)wave_tile * unit + 15u) & ~15u;

			if (stage_bytes < CR_UP2_ENTRIES_BYTES)
				stage_bytes = CR_UP2_ENTRIES_BYTES;   /* (k_up2 parks a wave-tile's converted window there first: UP2_ENTRIES_BYTES, cr_kup.hpp) */
			plan->lds_bytes = rows_bytes + (plan->threads / 64u) * (2u * 1024u + stage_bytes + 2u * CR_UP2_SLACK) + 16u;
			plan->tile_frames = (uint32_t)wave_tile * 4u;
			per_cu = (160u * 1024u) / plan->lds_bytes;
			if (per_cu > 2048u / plan->threads)
				per_cu = 2048u / plan->threads;
			if (per_cu >= 1u && plan->lds_bytes <= (uint32_t)di->max_lds_per_block)
			{
				plan->max_blocks = per_cu * (uint32_t)(di->compute_units > 0 ? di->compute_units : 256);
				return;
			}
		}

		if (g_env.debug)
			fprintf(stderr, "clownresampler_amd: k_up not used: increment %llu, ok %d, signs +%#x -%#x against mask %#x, wave tile %llu, lds %u of %d\n",
			        (unsigned long long)plan->increment, ok, pos_bits, neg_bits, negmask, (unsigned long long)wave_tile, plan->lds_bytes, di->max_lds_per_block);
		plan->variant = crhip_poly_up_fallback_variant(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode);
		crhip_poly_geometry(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode, plan->variant, &plan->threads, &plan->vecs, &frames_multiple);
	}

	if (plan->vecs >= 150u)
	{
		/* k_wave2 needs its slot signs, where it is built for fixed ones (checked like k_up's); geometry below with k_wave's */
		uint32_t negmask = 0, pos_bits = 0, neg_bits = 0;
		const int fixed = crhip_poly_wave2_negmask(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode, &negmask);
		int ok = fixed >= 0;

		if (fixed == 1)
		{
			const uint32_t safemask = crhip_poly_wave2_safemask(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode);

			cr_poly_slot_signs(&plan->poly, &pos_bits, &neg_bits);
			ok = (neg_bits & ~negmask) == 0 && (pos_bits & negmask) == 0;
			/* the mov-armed form stages every other slot's weights as |weight| << 15 */
			if (ok && safemask != 0 && (cr_poly_slots_reaching(&plan->poly, 65536) & ~safemask) != 0)
				ok = 0;
		}
		if (!ok)
		{
			plan->variant = crhip_poly_wave2_fallback_variant(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode);
			crhip_poly_geometry(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode, plan->variant, &plan->threads, &plan->vecs, &frames_multiple);
		}
	}

	if (plan->vecs >= 100u)
	{
		/* k_wave: every wave streams wave-tiles of 64 * ITER frames through a private, double-buffered (vecs - 100) KiB
		   slice of LDS; work is handed out in chunks of 4 wave-tiles (frames_multiple).  k_wave2 (vecs - 150 KiB pieces) keeps a
		   third buffer of twice the size beside them: the window expanded to one dword per sample. */
		const int wave2 = plan->vecs >= 150u;
		const uint32_t piece_bytes = (plan->vecs - (wave2 ? 150u : 100u)) * 1024u;
		const uint32_t per_wave = wave2 ? 4u * piece_bytes : 2u * piece_bytes;
		const uint32_t wave_tile = frames_multiple / 4u;
		const uint64_t last_rel = (65535u + (uint64_t)(wave_tile - 1u) * plan->increment) >> 16;
		const uint64_t window = 12u + (last_rel + window_slots) * frame_bytes;

		if (window <= piece_bytes && (uint64_t)wave_tile * plan->increment < (1ull << 32) - 65536u)
		{
			plan->lds_bytes = rows_bytes + (plan->threads / 64u) * per_wave + 16u; /* + the retired-waves counter */
			plan->tile_frames = frames_multiple;
			per_cu = (160u * 1024u) / plan->lds_bytes;
			if (per_cu > 2048u / plan->threads)
				per_cu = 2048u / plan->threads;
			if (per_cu >= 1u && plan->lds_bytes <= (uint32_t)di->max_lds_per_block)
			{
				plan->max_blocks = per_cu * (uint32_t)(di->compute_units > 0 ? di->compute_units : 256);
				return;
			}
		}

		/* the window of this configuration does not fit a wave's slice: use a k_poly variant instead */
		plan->variant = wave2 ? crhip_poly_wave2_fallback_variant(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode) : crhip_poly_fallback_variant();
		if (plan->variant >= 20u && plan->variant < 22u)
			plan->variant = crhip_poly_fallback_variant();
		crhip_poly_geometry(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode, plan->variant, &plan->threads, &plan->vecs, &frames_multiple);
	}

	tile_bytes = plan->vecs * 16u * plan->threads;
	plan->lds_bytes = rows_bytes + 2u * tile_bytes + 16u; /* + the ticket mailbox */

	if (plan->lds_bytes > (uint32_t)di->max_lds_per_block)
	{
		plan->use_poly = 0;
		plan->generic_reason = "polyphase rows + tiles exceed the LDS of one workgroup";
		return;
	}

	/* frames the tile image can hold after the (< 16 byte) alignment shift; 9-11 and 13-15 channels without a specialised instance repack
	   their tiles to frames of 32 bytes (the second buffer; the DMA buffer holds them as they are) */
	plan->padded = 0u;
	if (!plan->specialised && plan->vecs < 100u && crhip_poly_runtime_padded_frame_bytes(plan->channels) != 0u && plan->increment <= CR_PADDED_MAX_INCREMENT
	 && !g_env.no_padded_tiles)
	{
		plan->padded = 1u;
		frame_bytes = crhip_poly_runtime_padded_frame_bytes(plan->channels);
	}
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
	/* whole groups of threads * frames-in-flight; 4, 3, 2 or 1 groups per tile run as straight-line code in the kernel */
	{
		const int tile_groups = g_env.tile_groups; /* tuning hook (CLOWNRESAMPLER_AMD_TILE_GROUPS): cap the groups per tile */
		/* 13 and 15 channels (7-8 channels per lane plus the phantom channel): one group per tile.  With four groups - strong
		   upsampling - the straight-line tile measured 0.18 of the roofline against 0.30 (profiles/r01_channel_table.log);
		   at 44.1 <-> 48 kHz, where two groups fit, one costs 1-2 % */
		const uint64_t wide_phantom = (!plan->specialised && plan->channels > 12u && plan->channels % 2u == 1u) ? 1u : 0u;
		const uint64_t cap = tile_groups > 0 ? (uint64_t)tile_groups * frames_multiple : wide_phantom * frames_multiple;
		if (cap != 0 && tile > cap)
			tile = cap;
	}
	if (tile >= 4u * frames_multiple)
		tile = 4u * frames_multiple;
	else if (tile >= 3u * frames_multiple && plan->channels <= 2u)
		tile = 3u * frames_multiple;   /* (round 4: 48 -> 44.1 kHz fits three groups, not four: 2,048 -> 3,072-frame tiles; stereo + 4 %, and the
		                                  dual-mono launches on the stereo instance + 13 %; 4 and 8 channels measured 1-3 % SLOWER with three groups
		                                  and keep two: profiles/r04_three_group_tiles_ab.log) */
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
	plan->max_blocks = per_cu * (uint32_t)(di->compute_units > 0 ? di->compute_units : 256);
}

static int plan_key_matches(const ClownResamplerAMD_Plan *plan, uint64_t table_hash, unsigned radius, const cr_config *cfg, uint32_t channels, int device)
{
	return plan->table_hash == table_hash && plan->radius == radius && plan->channels == channels
	    && plan->key_variant == current_variant() && plan->device == device && memcmp(&plan->cfg, cfg, sizeof(*cfg)) == 0;
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
			plan_free(plan, g_ctx[plan->device] != NULL && g_ctx[plan->device]->ready);
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
	return cr_plan_get_on(current_device(), table_hash, table_len, fill_table, user, radius, cfg, channels, increment, pin);
}

/* The rotation of the rows in LDS that suits the plan's increment, for kernels that apply one (k_wave2 always; the 64-bit-chain
   k_poly has a plain and a rotated form, and the rotated one costs three instructions per frame: only where the model - extra LDS
   cycles per row read of a wave, cr_poly_pick_swizzle - gains CR_ROTATE_MIN_GAIN). */
/* k_wave2's lane order (crhip_poly_launch.lane_map): even / odd frames to the two halves of a wave where the model of the window
   reads' bank conflicts says that saves at least a quarter of a cycle per read. */
static uint32_t plan_pick_lane_map(const ClownResamplerAMD_Plan *plan)
{
	double straight, split;

	if (!(plan->vecs >= 150u && plan->vecs < 200u))
		return 0u;
	if (g_env.lane_map >= 0)
		return (uint32_t)g_env.lane_map;
	straight = cr_window_conflicts(&plan->poly, &plan->cfg, plan->increment, plan->channels, 0u);
	split = cr_window_conflicts(&plan->poly, &plan->cfg, plan->increment, plan->channels, 1u);
	if (g_env.debug)
		fprintf(stderr, "clownresampler_amd: plan %u ch, increment %llu: modelled conflict cycles per window read %.2f (lanes in frame order) / %.2f (even | odd)\n",
		        plan->channels, (unsigned long long)plan->increment, straight, split);
	return split < straight - 0.25 ? 1u : 0u;
}

static uint32_t plan_pick_rotation(const ClownResamplerAMD_Plan *plan, double *plain, double *best)
{
	const int form = crhip_poly_swizzled(plan->channels, plan->specialised ? plan->poly.slots : 0xFFFFu, plan->poly.row_mode, plan->poly.norm_mode, plan->variant);
	uint32_t rotation;

	*plain = *best = 0.0;
	if (form == 0)
		return 0u;
	rotation = cr_poly_pick_swizzle_mapped(&plan->poly, plan->increment, plan->lane_map, plain, best);
	/* (mono is the one shape whose LDS is busy enough for a smaller modelled gain to show: 44.1 -> 48 kHz models 12 -> 4 and measures
	   77.6 -> 75.9 us rotated, profiles/r04_mono_rotated_rows_ab.log, where stereo and wider frames measure 0.5-1 % slower) */
	if (form == 2 && *plain - *best < (g_env.rotate_min_gain >= 0.0 ? g_env.rotate_min_gain : (plan->channels == 1u ? 6.0 : CR_ROTATE_MIN_GAIN)))
		rotation = 0u;
	if (g_env.debug)
		fprintf(stderr, "clownresampler_amd: plan variant %u, increment %llu: rows rotated by %u (modelled conflict cycles per row read %.2f -> %.2f)\n",
		        plan->variant, (unsigned long long)plan->increment, rotation, *plain, *best);
	return rotation;
}

/* Once per plan, never inside a caller's stream capture: the kernel's function attributes, and the grid caps clamped to what is
   resident at once. */
/* `probe`: a shape the plan works without - a failure is reported to the caller only, never to the error handler (whose default aborts,
   and whose serial number callers compare). */
static int plan_prepare(const cr_device_ctx *ctx, ClownResamplerAMD_Plan *plan, int probe)
{
	crhip_poly_launch l;
	int form;

	fill_poly_launch(plan, &l);
	for (form = 0; form < 2; ++form)
	{
		const int code = crhip_poly_prepare(&l);

		if (code != 0 && probe)
			return -1;
		if (cr_check_hip(code, form ? "hipFuncSetAttribute(dynamic LDS, int16 form)" : "hipFuncSetAttribute(dynamic LDS)") != 0)
			return -1;
		l.out_s16 = 1; /* the int16-output instance is a different function */
	}

	/* The persistent grid must not be larger than what is resident at once: workgroups that start after the first
	   batch has left find their statically dealt tiles still waiting (twice the time) or no tickets (harmless).
	   LDS and thread count were accounted for above; registers are the runtime's to know - e.g. above 96 SGPRs a
	   SIMD holds 7 waves, not 8, and a 1024-thread workgroup then has the CU to itself. */
	plan->max_blocks_s16 = plan->max_blocks;
	for (form = 0; form < 2; ++form)
	{
		int per_cu = 0, vgprs = 0, static_lds = 0;
		l.out_s16 = (uint32_t)form;
		if (crhip_poly_occupancy(&l, &per_cu, &vgprs, &static_lds) == 0 && per_cu >= 1)
		{
			const uint32_t resident = (uint32_t)per_cu * (uint32_t)(ctx->info.compute_units > 0 ? ctx->info.compute_units : 256);
			if (g_env.debug)
				fprintf(stderr, "clownresampler_amd: plan variant %u (%s output): %u threads, %u B dynamic LDS, %d VGPRs: %d workgroups per CU, grid cap %u -> %u\n",
				        plan->variant, form ? "int16" : "int32", plan->threads, plan->lds_bytes, vgprs, per_cu, form ? plan->max_blocks_s16 : plan->max_blocks, resident);
			if (g_env.no_occupancy_clamp) /* (tuning hook CLOWNRESAMPLER_AMD_NO_OCCUPANCY_CLAMP: measure without) */
				continue;
			if (form == 0 && resident < plan->max_blocks)
				plan->max_blocks = resident;
			if (form == 1 && resident < plan->max_blocks_s16)
				plan->max_blocks_s16 = resident;
		}
	}
	return 0;
}

/* DUAL MONO: a mono plan's private stereo partner over the mono plan's OWN rows (the image of a configuration's rows does not
   depend on the channel count where both instances take the same layout), if the stereo instance of this configuration has a dual
   form.  Nothing is reported when there is none: the mono kernels then do all the work, as before. */
static void plan_dual_partner(const cr_device_ctx *ctx, ClownResamplerAMD_Plan *plan)
{
	ClownResamplerAMD_Plan *partner;
	crhip_poly_launch l;
	uint64_t g = plan->increment, period = 65536u;
	int per_cu = 0, vgprs = 0, static_lds = 0;

	plan->dual.partner = NULL;
	if (plan->channels != 1u || !plan->use_poly || g_env.no_dual_mono || g_force_generic)
		return;
	while (period > 1u && (g & 1u) == 0u)   /* 65536 / gcd(increment, 65536) */
	{
		g >>= 1;
		period >>= 1;
	}

	partner = (ClownResamplerAMD_Plan *)calloc(1, sizeof(*partner));
	if (partner == NULL)
		return;
	*partner = *plan;
	partner->next = NULL;
	partner->channels = 2u;
	partner->variant = partner->key_variant;
	memset(&partner->brief, 0, sizeof(partner->brief));
	memset(&partner->intk, 0, sizeof(partner->intk));
	memset(&partner->dual, 0, sizeof(partner->dual));
	plan_geometry(partner);
	if (partner->use_poly && partner->vecs >= 200u && partner->key_variant == (uint32_t)CR_DEFAULT_VARIANT)
	{
		/* the stereo instance's kernel at this ratio is k_up2 (8x - 13x upsampling with 8 lobes: the mono version of cfg 3), which has no
		   dual form: the partner takes the instance's other kernel (k_wave2), as the plan's own brief launches do */
		partner->variant = crhip_poly_up_fallback_variant(partner->channels, partner->poly.slots, partner->poly.row_mode, partner->poly.norm_mode);
		plan_geometry(partner);
	}
	partner->lane_map = plan_pick_lane_map(partner);
	partner->lds_swizzle = plan_pick_rotation(partner, &partner->conflict_plain, &partner->conflict_best);
	/* the stereo instance must be a specialised k_poly over the same image layout, with a dual form, whose two mono windows fit
	   the halves of its DMA buffer (a stereo window of W frames fits the whole: W * 4 + 12 <= bytes; each mono one needs W * 2 + 14) */
	fill_poly_launch(partner, &l);
	l.dual = 1u;
	{
		/* frames of the longest tile's window, and the bytes ONE mono window may take: half of the stereo instance's DMA buffer
		   (k_poly: vecs * 16 bytes per thread; k_wave2: vecs - 150 KiB per wave-tile of 64 * ITER frames = tile_frames / 4) */
		const int wave2 = partner->vecs >= 150u && partner->vecs < 200u;
		const uint64_t tile = wave2 ? partner->tile_frames / 4u : partner->tile_frames;
		const uint64_t window = (((uint64_t)65535u + (tile - 1u) * plan->increment) >> 16) + plan->poly.slots + plan->poly.window_extra;
		const uint64_t half_bytes = wave2 ? (uint64_t)(partner->vecs - 150u) * 1024u / 2u : (uint64_t)partner->vecs * 16u * partner->threads / 2u;

		if (!partner->use_poly || (partner->vecs >= 100u && !wave2) || !partner->specialised || !plan->specialised
		 || plan_image_stride(partner) != plan->device_row_stride || !crhip_poly_has_dual(&l)
		 || window * 2u + 4u > half_bytes   /* (the dual fetches start at the 4-byte word of a window's first sample) */
		 || crhip_poly_prepare(&l) != 0)
		{
			free(partner);
			return;
		}
	}
	partner->max_blocks_s16 = partner->max_blocks;
	if (crhip_poly_occupancy(&l, &per_cu, &vgprs, &static_lds) == 0 && per_cu >= 1 && !g_env.no_occupancy_clamp)
	{
		const uint32_t resident = (uint32_t)per_cu * (uint32_t)(ctx->info.compute_units > 0 ? ctx->info.compute_units : 256);

		if (resident < partner->max_blocks)
			partner->max_blocks = resident;
	}
	if (g_env.debug)
		fprintf(stderr, "clownresampler_amd: mono plan, increment %llu: dual-mono partner on the stereo instance (variant %u, tile %u frames, %u B LDS, grid cap %u, period %llu)\n",
		        (unsigned long long)plan->increment, partner->variant, partner->tile_frames, partner->lds_bytes, partner->max_blocks, (unsigned long long)period);
	plan->store->refs += 1;   /* (the partner's view of the rows) */
	plan->dual.partner = partner;
	plan->dual.period = period;
	plan->dual.max_blocks = partner->max_blocks;
}

/* The shape of k_up's fallback kernel beside a k_up plan's own, for its brief launches (see the plan's `brief`). */
static void plan_brief_shape(const cr_device_ctx *ctx, ClownResamplerAMD_Plan *plan)
{
	ClownResamplerAMD_Plan other;
	double plain, best;
	const uint32_t half_tiles = g_env.brief_half_tiles >= 0 ? (uint32_t)g_env.brief_half_tiles
	                          : (plan->poly.slots >= 15u ? CR_BRIEF_HALF_TILES_LONG_WINDOWS : CR_BRIEF_HALF_TILES);

	plan->brief.below = 0;
	if (plan->vecs < 200u || plan->key_variant != (uint32_t)CR_DEFAULT_VARIANT || half_tiles == 0)
		return;

	other = *plan;   /* (a scratch copy: nothing in it is owned) */
	other.variant = crhip_poly_up_fallback_variant(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode);
	plan_geometry(&other);
	if (!other.use_poly || other.vecs >= 200u || other.specialised != plan->specialised)
		return;   /* (the fallback shape does not fit this plan's rows, or is no other kernel) */
	other.lane_map = plan_pick_lane_map(&other);
	other.lds_swizzle = plan_pick_rotation(&other, &plain, &best);
	if (plan_prepare(ctx, &other, 1) != 0)
		return;   /* (the plan works without) */

	plan->brief.threads = other.threads;
	plan->brief.vecs = other.vecs;
	plan->brief.tile_frames = other.tile_frames;
	plan->brief.lds_bytes = other.lds_bytes;
	plan->brief.max_blocks = other.max_blocks;
	plan->brief.max_blocks_s16 = other.max_blocks_s16;
	plan->brief.variant = other.variant;
	plan->brief.lds_swizzle = other.lds_swizzle;
	plan->brief.lane_map = other.lane_map;
	plan->brief.below = (uint64_t)half_tiles * (plan->tile_frames / 4u) * plan->max_blocks * (plan->threads / 64u) / 2u;
}

/* k_int beside the plan's ordinary kernel: see the plan's `intk`. */
/* k_seg for this plan?  The shape k_up2 is the kernel of (the instance's slot signs hold for every row), at the ratios where a tile's
   window fits its registers and LDS entries; the float image of the rows is made once per store.  0 on success (available or not). */
static int plan_seg_shape(const cr_device_ctx *ctx, ClownResamplerAMD_Plan *plan)
{
	uint32_t negmask = 0, pos_bits = 0, neg_bits = 0, threads = 0, lds = 0, chunk = 0;
	cr_plan_store *store = plan->store;
	int per_cu = 0;
	uint64_t g;

	(void)ctx;
	plan->seg.available = 0;
	if (g_env.no_seg || g_env.no_special || !plan->use_poly || plan->poly.row_mode != CRHIP_ROWMODE_UPSAMPLE || plan->poly.window_extra != 0 || plan->increment >= 65536u
	 || !crhip_seg_instance(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode, (uint32_t)plan->increment, &negmask, &threads, &lds, &chunk))
		return 0;
	cr_poly_slot_signs(&plan->poly, &pos_bits, &neg_bits);
	if ((neg_bits & ~negmask) != 0 || (pos_bits & negmask) != 0 || cr_poly_slots_reaching(&plan->poly, 65537) != 0 || plan->poly.rows > 1025u)
		return 0;
	if (plan->increment < CR_SEG_MIN_INCREMENT)
		return 0;

	if (store->d_rows_seg == NULL)
	{
		const size_t bytes = (size_t)plan->poly.rows * 64u;
		uint32_t *image = (uint32_t *)malloc(bytes);
		uint32_t r, s;
		int failed;

		if (image == NULL)
			return cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "out of host memory");
		for (r = 0; r < plan->poly.rows; ++r)
		{
			const int32_t *row = plan->poly.weights + (size_t)r * plan->poly.row_stride;

			for (s = 0; s < 15u; ++s)
			{
				/* |weight| <= 65536: exact as a float, and so is the division by 2^16 */
				const float f = (float)(row[s] < 0 ? -(int64_t)row[s] : (int64_t)row[s]) * (1.0f / 65536.0f);
				memcpy(&image[r * 16u + s], &f, sizeof(f));
			}
			image[r * 16u + 15u] = 2u * (uint32_t)row[plan->poly.slots];   /* the reciprocal, doubled: k_up2's normalisation (cr_kup.hpp) */
		}
		/* k_seg is an OPTIONAL fast path: whatever fails here leaves the plan without it (k_up2 / k_wave2 serve the configuration), and the
		   store only ever holds an image that has arrived whole - a sibling plan must not find a pointer to rows nobody filled */
		void *rows = NULL;

		failed = crhip_malloc(&rows, bytes) != 0;
		flight_memory(CR_FLIGHT_MALLOC, failed ? NULL : rows, bytes, failed);
		if (!failed && (copy_pageable_h2d(rows, image, bytes, NULL) != 0 || crhip_stream_sync(NULL) != 0))
		{
			dev_free(rows);
			failed = 1;
		}
		free(image);
		if (failed)
			return 0;
		store->d_rows_seg = rows;
	}

	if (crhip_seg_prepare(plan->channels, plan->poly.slots, (uint32_t)plan->increment, &per_cu) != 0 || per_cu < 1)
		return 0;
	/* 65536 / gcd(increment, 65536) */
	for (g = 65536u; g > 1u && (plan->increment & (65536u / g * 2u - 1u)) == 0; g >>= 1)
		;
	plan->seg.period = g;
	plan->seg.threads = threads;
	plan->seg.chunk = chunk;
	plan->seg.lds_bytes = lds;
	plan->seg.max_blocks = (uint32_t)per_cu * (uint32_t)(g_ctx[plan->device]->info.compute_units > 0 ? g_ctx[plan->device]->info.compute_units : 256);
	plan->seg.d_rows = store->d_rows_seg;
	plan->seg.available = 1;
	return 0;
}

static void plan_int_shape(const cr_device_ctx *ctx, ClownResamplerAMD_Plan *plan)
{
	int per_cu = 0, per_cu_s16 = 0;
	uint32_t period;
	const uint32_t cus = (uint32_t)(ctx->info.compute_units > 0 ? ctx->info.compute_units : 256);

	plan->intk.available = 0;
	if (g_env.no_int_kernel || !plan->use_poly || plan->increment == 0 || plan->key_variant != (uint32_t)CR_DEFAULT_VARIANT)
		return;
	/* the period of the fractional position: 1 (a whole-number ratio), 2 or 4 */
	for (period = 1; period <= 4u; period *= 2u)
		if (((plan->increment * period) & 0xFFFFu) == 0)
			break;
	if (period > 4u || (plan->increment * period) >> 16 > 64u || plan->poly.slots * period > CRHIP_INT_MAX_SLOTS
	 || (period == 1u && plan->poly.row_mode != CRHIP_ROWMODE_AFFINE))
		return;
	plan->intk.period = period;
	plan->intk.ratio = (uint32_t)((plan->increment * period) >> 16);
	if (!crhip_int_instance(plan->channels, plan->intk.ratio, period, plan->poly.slots, &plan->intk.shape))
		return;
	if (crhip_int_prepare(plan->channels, plan->intk.ratio, period, plan->poly.slots, &per_cu, &per_cu_s16) != 0 || per_cu < 1 || per_cu_s16 < 1)
		return;   /* (the plan works without) */
	plan->intk.max_blocks = (uint32_t)per_cu * cus;
	plan->intk.max_blocks_s16 = (uint32_t)per_cu_s16 * cus;
	plan->intk.available = 1;
	if (g_env.debug)
		fprintf(stderr, "clownresampler_amd: plan %u ch, ratio %u:%u, %u slots: k_int with %u frames per lane, %d workgroups of %u threads per CU\n",
		        plan->channels, plan->intk.ratio, plan->intk.period, plan->poly.slots, plan->intk.shape.frames_per_lane, per_cu, plan->intk.shape.threads);
}

/* The staged rows of a k_int launch from this fractional position on (one per phase of the period), or 0 when they do not have
   the instance's slot classes or window starts. */
static int int_launch_row(const ClownResamplerAMD_Plan *plan, uint32_t frac, crhip_int_launch *l, uint32_t *first_slot)
{
	const cr_poly *poly = &plan->poly;
	const uint32_t period = plan->intk.period;
	uint32_t rows[4], starts[4], p, s;

	if (poly->weights == NULL || !cr_poly_periodic(poly, plan->increment, frac, period, rows, starts))
		return 0;
	for (p = 0; p < period; ++p)
	{
		const int32_t *w = poly->weights + (size_t)rows[p] * poly->row_stride;

		if (starts[p] - starts[0] != plan->intk.shape.starts[p])
			return 0;
		for (s = 0; s < poly->slots; ++s)
		{
			const uint32_t at = p * poly->slots + s;
			const int negative = (int)((plan->intk.shape.negmask >> at) & 1u);
			const int safe = (int)((plan->intk.shape.safemask >> at) & 1u);
			const int64_t magnitude = negative ? -(int64_t)w[s] : (int64_t)w[s];

			if (magnitude < 0 || magnitude > 65536 || (!safe && magnitude == 65536) || (((plan->intk.shape.zeromask >> at) & 1u) && magnitude != 0))
				return 0;
			l->w[at] = safe ? (int32_t)magnitude : (int32_t)((uint32_t)magnitude << 15);
		}
		l->reciprocal[p] = w[poly->slots];
	}
	/* the frame slot 0 of the launch's first output frame multiplies, counted from its position_integer */
	*first_slot = starts[0];
	return 1;
}

ClownResamplerAMD_Plan *cr_plan_get_on(int device, uint64_t table_hash, size_t table_len, cr_table_fill fill_table, const void *user,
                                       unsigned radius, const cr_config *cfg, uint32_t channels, uint64_t increment, int pin)
{
	ClownResamplerAMD_Plan *plan, *sibling = NULL;
	cr_plan_store *store = NULL;
	int32_t *table = NULL;
	cr_device_ctx *ctx;

	pthread_mutex_lock(&g_lock);

	ctx = ensure_ctx_locked(device);
	if (ctx == NULL)
		goto fail;

	for (plan = g_plans; plan != NULL; plan = plan->next)
	{
		if (!plan_key_matches(plan, table_hash, radius, cfg, channels, device))
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
	plan->device = device;
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

		if (cr_check_hip(dev_malloc((void **)&store->d_table, table_len * sizeof(int32_t)), "hipMalloc(table)") != 0
		 || cr_check_hip(copy_pageable_h2d(store->d_table, table, table_len * sizeof(int32_t), NULL), "hipMemcpy(table)") != 0
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

	/* instances that rotate the rows while staging them into LDS get the rotation that suits THIS plan's increment (the image
	   in global memory stays the plain one, shared by the plans of every increment) */
	plan->lane_map = plan->use_poly ? plan_pick_lane_map(plan) : 0u;
	plan->lds_swizzle = plan->use_poly ? plan_pick_rotation(plan, &plan->conflict_plain, &plan->conflict_best) : 0u;

	if (plan->use_poly)
	{
		const int layout = plan->specialised ? CR_IMAGE_COMPACT : CR_IMAGE_SPLIT;

		plan->swizzle = 0;   /* (the image in global memory is never swizzled: see lds_swizzle) */

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

			if (cr_check_hip(dev_malloc((void **)&store->d_rows, bytes), "hipMalloc(rows)") != 0
			 || cr_check_hip(copy_pageable_h2d(store->d_rows, image, bytes, NULL), "hipMemcpy(rows)") != 0
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

		if (plan_prepare(ctx, plan, 0) != 0)
			goto fail_plan;
		plan_brief_shape(ctx, plan);
		plan_int_shape(ctx, plan);
		plan_dual_partner(ctx, plan);
		if (plan_seg_shape(ctx, plan) != 0)
			goto fail_plan;
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
	l->debug_form = (uint32_t)g_env.w2_form;
	l->plane_rows = plan->plane_rows;
	l->swizzle = plan->lds_swizzle;
	l->padded = plan->padded;
	l->lane_map = plan->lane_map;
	l->wave_tile = plan->wave_tile;
	l->debug_stamps = g_debug_stamps;
}

/* Whether a dual-mono launch of n_out mono frames, split at `half`, stays inside the 32-bit arithmetic of the kernels' buffer descriptors:
   the dual forms of k_poly / k_wave2 store both halves through ONE descriptor over the mono output (4 bytes per frame, clamped to
   0xFFFFFFFC bytes) with 32-bit byte offsets that run up to a tile past the last frame, and fetch both windows through ONE descriptor from the
   first window's start to the end of the caller's buffer (also clamped).  Beyond that a store would be dropped or an offset would wrap - the
   ordinary mono kernels (64-bit pointers) take such launches. */
int cr_dual_mono_fits(uint64_t n_out, uint64_t half, uint64_t tile_frames, uint64_t increment, uint64_t in_valid_bytes)
{
	const uint64_t limit = 0xFFFFFFFCull;

	if (half >= (1ull << 30) || ((half * increment) >> 16) >= (1ull << 31))
		return 0;
	if (n_out * 4u + 4u * tile_frames * 4u > limit)   /* the output, plus the ragged last tile's offsets */
		return 0;
	if (in_valid_bytes > limit)                        /* the input: both windows inside one descriptor */
		return 0;
	return 1;
}

int ClownResamplerAMD_DebugDualMonoFits(uint64_t n_out, uint64_t half, uint64_t tile_frames, uint64_t increment, uint64_t in_valid_bytes)
{
	return cr_dual_mono_fits(n_out, half, tile_frames, increment, in_valid_bytes);
}

int cr_plan_launch(const ClownResamplerAMD_Plan *plan, const void *d_in, uint64_t in_valid_bytes, void *d_out,
                   uint64_t pos_int, uint64_t pos_frac, uint64_t n_out, void *stream, int out_s16)
{
	if (n_out == 0)
		return 0;

	if (plan->use_poly && !g_force_generic && !g_no_int_kernel && plan->intk.available && pos_int < (1ull << 47) && n_out < (1ull << 40))
	{
		/* a whole-number ratio: k_int, if the row this launch's fraction selects has the instance's slot classes */
		crhip_int_launch il;
		uint32_t first_slot = 0;

		memset(&il, 0, sizeof(il));
		/* A periodic ratio's instance is compiled for ONE order of the phases.  A long launch that starts elsewhere in the period - the
		   piece before it ended after an odd number of frames - hands its first one to three frames to the plan's ordinary kernel and
		   starts k_int at the phase the instance begins with (same stream: the two launches write disjoint frames). */
		if (plan->intk.period > 1u && n_out >= CR_INT_SPLIT_MIN_FRAMES && !int_launch_row(plan, (uint32_t)pos_frac, &il, &first_slot))
		{
			uint32_t j;

			for (j = 1; j < plan->intk.period; ++j)
			{
				const uint64_t pos = pos_frac + (uint64_t)j * plan->increment;

				if (int_launch_row(plan, (uint32_t)(pos & 0xFFFFu), &il, &first_slot))
				{
					const size_t frame_bytes = (size_t)plan->channels * (out_s16 ? 2u : 4u);

					if (cr_plan_launch(plan, d_in, in_valid_bytes, d_out, pos_int, pos_frac, j, stream, out_s16) != 0)
						return -1;
					return cr_plan_launch(plan, d_in, in_valid_bytes, (unsigned char *)d_out + (size_t)j * frame_bytes, pos_int + (pos >> 16), pos & 0xFFFFu,
					                      n_out - j, stream, out_s16);
				}
			}
		}
		if (int_launch_row(plan, (uint32_t)pos_frac, &il, &first_slot))
		{
			const uint64_t tile = 64ull * plan->intk.shape.frames_per_lane;   /* (output frames) */
			const uint64_t waves = (n_out + tile - 1) / tile;
			const uint32_t waves_per_block = plan->intk.shape.threads / 64u;
			const uint32_t cap = out_s16 ? plan->intk.max_blocks_s16 : plan->intk.max_blocks;
			uint64_t blocks = (waves + waves_per_block - 1) / waves_per_block;

			il.d_in = d_in;
			il.in_valid_bytes = in_valid_bytes;
			il.d_out = d_out;
			il.first_frame = pos_int + first_slot;
			il.n_out = n_out;
			il.channels = plan->channels;
			il.ratio = plan->intk.ratio;
			il.period = plan->intk.period;
			il.slots = plan->poly.slots;
			il.out_s16 = out_s16 ? 1u : 0u;
			il.blocks = (uint32_t)(blocks > cap ? cap : blocks);
			__atomic_fetch_add(&g_launch_count[5], 1ull, __ATOMIC_RELAXED);
			/* tickets where a wave has several tiles to take (a draw is a memory round trip per tile: short launches are dealt statically) */
			if (g_env.dynamic_tiles != 0 && waves >= 4ull * il.blocks * waves_per_block)
			{
				int ring, e;

				il.d_tickets = ticket_block_for(g_ctx[plan->device], stream, &ring);
				if (il.d_tickets == NULL)
					return -1;
				/* tiles per ticket: every wave still gets ~8 draws, and the 32 counters are not hammered by launches of short tiles */
				{
					const uint64_t per_wave = waves / ((uint64_t)il.blocks * waves_per_block);
					il.ticket_tiles = per_wave >= 32u ? 4u : (per_wave >= 16u ? 2u : 1u);
				}
				e = launch_int(plan, &il, stream);
				ticket_block_enqueued(g_ctx[plan->device], ring);
				__atomic_fetch_add(&g_launch_count[CR_COUNT_TICKETED], 1ull, __ATOMIC_RELAXED);
				return cr_check_hip(e, "k_int launch");
			}
			return cr_check_hip(launch_int(plan, &il, stream), "k_int launch");
		}
	}

	if (plan->seg.available && !out_s16 && !g_force_generic && g_seg_mode != 2 && pos_int < (1ull << 40) && n_out < (1ull << 40) && in_valid_bytes < 0xFFFFFFFCull
	 && (g_variant < 0 || g_variant == CR_DEFAULT_VARIANT))
	{
		/* k_seg: the lanes of a wave S output frames apart, S a multiple of the fraction's period (equal fractions: one row per wave and
		   step, in scalar registers).  S is the smallest such multiple of 1,024 frames or more (segments of a few tiles), a super-block 64 S; the lanes of
		   the last super-block that lie beyond the launch idle, and a launch that would waste more than CR_SEG_MAX_WASTE of its
		   lane-steps that way stays with k_up2 (cfg 3, odd increment: S = 65536, 13.7 super-blocks in ten minutes, 1.9 %). */
		const uint64_t period = plan->seg.period;
		const uint64_t seg = period >= 1024u ? period : 1024u;   /* (both powers of two) */
		const uint64_t blocks64 = (n_out + 64u * seg - 1u) / (64u * seg);
		const double waste = 1.0 - (double)n_out / ((double)blocks64 * 64.0 * (double)seg);
		/* Frames per lane and tile (a multiple of the chunk; a segment's last tile may be shorter): 128 while that leaves the waves two
		   tiles each or more, else 64.  Measured on cfg 3 (profiles/r05_kseg_tiles2.log): 64: 118.5 us, 128: 116.9, 144: 121.0, 160: 121.3
		   (5,740 tiles on 3,072 waves: two each at most - and slower: a wave that has finished leaves its SIMD to the others, so many
		   short tiles balance better than few long ones), 176: 130.4, 192: 136.2; a tile's first window (scattered loads, 90 conversions
		   per lane, a drained store queue) is what speaks against 64. */
		uint64_t tile = CR_SEG_MAX_TILE;
		const uint64_t waves = (uint64_t)plan->seg.max_blocks * (plan->seg.threads / 64u);

		while (tile > CR_SEG_MIN_TILE && blocks64 * ((seg + tile - 1u) / tile) < 2u * waves)
			tile /= 2u;
		if (g_env.seg_tile >= 16 && (uint64_t)g_env.seg_tile <= seg)
			tile = (uint64_t)g_env.seg_tile;
		if (tile % plan->seg.chunk == 0 && seg * 512u < (1ull << 32) && ((seg * plan->increment) >> 16) * 256u < (1ull << 32)
		 && blocks64 * ((seg + tile - 1u) / tile) < (1ull << 32)
		 && (g_seg_mode == 1 || (waste <= CR_SEG_MAX_WASTE && blocks64 * ((seg + tile - 1u) / tile) >= waves + waves / 2u)))
		{
			crhip_seg_launch sl;
			int ring, e;
			uint64_t grid;

			memset(&sl, 0, sizeof(sl));
			sl.d_in = d_in;
			sl.in_valid_bytes = in_valid_bytes;
			sl.d_out = d_out;
			sl.d_rows = plan->seg.d_rows;
			sl.pos0 = (pos_int << 16) + pos_frac;
			sl.n_out = n_out;
			sl.seg_frames = seg;
			sl.seg_in_frames = (seg * plan->increment) >> 16;   /* (exact: seg is a multiple of the period) */
			sl.increment = (uint32_t)plan->increment;
			sl.first_slot = plan->poly.first_slot;
			sl.slots = plan->poly.slots;
			sl.tile_frames = (uint32_t)tile;
			sl.tiles_per_seg = (uint32_t)((seg + tile - 1u) / tile);
			sl.n_tiles = blocks64 * sl.tiles_per_seg;
			sl.debug_form = (uint32_t)g_env.seg_form;
			sl.debug_stamps = g_debug_stamps;
			sl.xcd_run = (uint32_t)g_env.seg_xcd_run;
			grid = (sl.n_tiles + plan->seg.threads / 64u - 1u) / (plan->seg.threads / 64u);
			sl.blocks = (uint32_t)(grid > plan->seg.max_blocks ? plan->seg.max_blocks : grid);
			sl.d_tickets = ticket_block_for(g_ctx[plan->device], stream, &ring);
			if (sl.d_tickets == NULL)
				return -1;
			e = launch_seg(plan, &sl, stream);
			ticket_block_enqueued(g_ctx[plan->device], ring);
			__atomic_fetch_add(&g_launch_count[CR_COUNT_SEG], 1ull, __ATOMIC_RELAXED);
			return cr_check_hip(e, "k_seg launch");
		}
	}

	if (plan->dual.partner != NULL && !out_s16 && !g_force_generic && !g_no_dual_mono && pos_int < (1ull << 47) && n_out < (1ull << 40))
	{
		/* DUAL MONO: a long mono launch on the stereo instance - output frames j and j + H as its two channels.  H = half the launch,
		   rounded UP to a multiple of the period of the fraction (equal fractions: one row for both):
		   the second half is the shorter one, its missing frames are computed on zeros / neighbours and not stored - less than one part
		   in sixteen of the launch by the rule below, 0.1 % for ten minutes of audio. */
		const ClownResamplerAMD_Plan *partner = plan->dual.partner;
		/* (H a multiple of the period only: the last tile of the pairs may be a ragged one - round 4's first form rounded H to whole
		   tiles too, lcm(65536, 3072) = 196,608 frames for the 3-lobe instances, and took launches from 3.1 M output frames on) */
		const uint64_t unit = plan->dual.period;
		const uint64_t half = ((n_out + 1u) / 2u + unit - 1u) / unit * unit;

		if (n_out >= 16u * unit && half >= 8ull * partner->tile_frames && half < n_out && cr_dual_mono_fits(n_out, half, partner->tile_frames, plan->increment, in_valid_bytes))
		{
			crhip_poly_launch l;
			uint64_t blocks;
			const int wave2 = partner->vecs >= 150u;
			int ring, e;

			fill_poly_launch(partner, &l);
			if (wave2)
			{
				/* (k_wave2's chunks: see the ordinary launch below) */
				const uint32_t wave_tile = partner->tile_frames / 4u;
				const uint64_t waves = (uint64_t)plan->dual.max_blocks * (partner->threads / 64u);

				while (l.tile_frames > wave_tile && half / l.tile_frames < 8u * waves)
					l.tile_frames /= 2u;
				blocks = (half / l.tile_frames + partner->threads / 64u - 1u) / (partner->threads / 64u);
			}
			else
				blocks = (half + partner->tile_frames - 1u) / partner->tile_frames;
			l.d_in = d_in;
			l.in_valid_bytes = in_valid_bytes;
			l.d_out = d_out;
			l.pos0 = (pos_int << 16) + pos_frac;
			l.n_out = half;
			l.dual = 1u;
			l.dual_out_frames = (uint32_t)half;
			l.dual_valid_frames = (uint32_t)(n_out - half);
			l.dual_in_bytes = (uint32_t)(((half * plan->increment) >> 16) * 2u);   /* (half * increment is a multiple of 65536; < 2^32: checked above) */
			if (blocks > plan->dual.max_blocks)
				blocks = plan->dual.max_blocks;
			l.blocks = (uint32_t)blocks;
			l.dynamic_tiles = g_env.dynamic_tiles >= 0 ? (uint32_t)g_env.dynamic_tiles
			                                           : (uint32_t)crhip_poly_dynamic_default(2u, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode);
			if (g_env.dynamic_tiles < 0 && half / partner->tile_frames < 8ull * blocks)
				l.dynamic_tiles = 0u;
			l.d_tickets = ticket_block_for(g_ctx[plan->device], stream, &ring);
			if (l.d_tickets == NULL)
				return -1;
			e = launch_poly(plan, &l, stream);
			ticket_block_enqueued(g_ctx[plan->device], ring);
			__atomic_fetch_add(&g_launch_count[wave2 ? 4 : 1], 1ull, __ATOMIC_RELAXED);
			if (!wave2 && l.dynamic_tiles != 0u)
				__atomic_fetch_add(&g_launch_count[CR_COUNT_TICKETED], 1ull, __ATOMIC_RELAXED);
			return cr_check_hip(e, "k_poly launch (dual mono)");
		}
	}

	if (plan->use_poly && !g_force_generic && pos_int < (1ull << 47) && n_out < (1ull << 40))
	{
		crhip_poly_launch l;
		uint64_t blocks;
		uint32_t threads = plan->threads, vecs = plan->vecs, shape_tile = plan->tile_frames;
		uint32_t max_blocks = out_s16 ? plan->max_blocks_s16 : plan->max_blocks;

		fill_poly_launch(plan, &l);
		l.d_in = d_in;
		l.in_valid_bytes = in_valid_bytes;
		l.d_out = d_out;
		l.pos0 = (pos_int << 16) + pos_frac;
		l.n_out = n_out;
		l.out_s16 = out_s16 ? 1u : 0u;
		if (n_out < plan->brief.below)
		{
			/* a brief launch of a k_up plan: the instance's other kernel, over the same rows (see the plan's `brief`) */
			l.threads = threads = plan->brief.threads;
			l.vecs = vecs = plan->brief.vecs;
			l.tile_frames = shape_tile = plan->brief.tile_frames;
			l.lds_bytes = plan->brief.lds_bytes;
			l.variant = plan->brief.variant;
			l.swizzle = plan->brief.lds_swizzle;
			l.lane_map = plan->brief.lane_map;
			max_blocks = out_s16 ? plan->brief.max_blocks_s16 : plan->brief.max_blocks;
		}

		if (vecs >= 150u && vecs < 200u)
		{
			/* k_wave2 draws chunks of 4 wave-tiles; a launch that leaves a wave only two or three of those ends with a third of
			   the waves idle, so short launches get chunks of 2 or 1 (the kernel takes the chunk size from the launch) */
			const uint32_t wave_tile = shape_tile / 4u;
			const uint64_t waves = (uint64_t)max_blocks * (threads / 64u);

			while (l.tile_frames > wave_tile && n_out / l.tile_frames < 8u * waves)
				l.tile_frames /= 2u;
		}

		/* tiles are dealt round-robin to a persistent grid (see k_poly) */
		blocks = (n_out + l.tile_frames - 1) / l.tile_frames;
		if (vecs >= 100u)
			blocks = (blocks + threads / 64u - 1) / (threads / 64u); /* k_wave hands chunks to WAVES */
		if (blocks > max_blocks)
			blocks = max_blocks;
		l.blocks = (uint32_t)blocks;
		{
			/* tickets pay off where workgroups drift apart over many medium-sized tiles; measured on MI355X (profiles/):
			   stereo 3-lobe upsampling gains ~4 %, 8-channel and 8-lobe instances lose 1-8 %: a per-instance default */
			l.dynamic_tiles = g_env.dynamic_tiles >= 0 ? (uint32_t)g_env.dynamic_tiles
			                                           : (uint32_t)crhip_poly_dynamic_default(plan->channels, plan->poly.slots, plan->poly.row_mode, plan->poly.norm_mode);
			/* ... and, whatever the instance, LONG launches of wide frames: with 8 channels and more a tile is 512-1024 frames, a
			   10-minute stream is a few hundred tiles per workgroup, and the workgroups drift apart (bench.py --workload up8 /
			   up12, 10 minutes: 253 -> 239 us, 505 -> 460 us with tickets; the channel table's launches of ~40 tiles per
			   workgroup: within +-2 % either way) */
			/* ... and never for SHORT launches: a draw is a memory round trip in wave 0 (the other waves wait for it at the tile's
			   barrier), one in the prologue, one per tile, and a count-down at the end.  With a dozen tiles per workgroup the
			   second workgroup of a CU hides that and the balance is worth 1 %; with one or two it is most of what the launch
			   spends outside its tiles (tools/size_sweep.py, stereo 44.1 -> 48 kHz: 30 s 9.5 -> 7.6 us, one minute 12.3 -> 9.0 us,
			   two minutes 16.1 -> 14.5 us, five minutes 29.5 -> 29.3 us, ten minutes 61.3 against 62.0 us the other way) */
			if (g_env.dynamic_tiles < 0 && vecs < 100u && (n_out + l.tile_frames - 1) / l.tile_frames < 8ull * blocks)
				l.dynamic_tiles = 0u;
			if (g_env.dynamic_tiles < 0 && plan->channels >= 8u && vecs < 100u
			 && (n_out + l.tile_frames - 1) / l.tile_frames >= 48ull * max_blocks)
				l.dynamic_tiles = 1u;
		}
		{
			int ring, e;

			l.d_tickets = ticket_block_for(g_ctx[plan->device], stream, &ring);
			if (l.d_tickets == NULL)
				return -1;
			e = launch_poly(plan, &l, stream);
			ticket_block_enqueued(g_ctx[plan->device], ring);
			__atomic_fetch_add(&g_launch_count[l.variant == CRHIP_VARIANT_RT_WAVE2S ? 6 : vecs >= 200u ? 3 : vecs >= 150u ? 4 : vecs >= 100u ? 2 : 1], 1ull, __ATOMIC_RELAXED);
			if (vecs < 100u && l.dynamic_tiles != 0u)
				__atomic_fetch_add(&g_launch_count[CR_COUNT_TICKETED], 1ull, __ATOMIC_RELAXED);
			return cr_check_hip(e, "k_poly launch");
		}
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

		__atomic_fetch_add(&g_launch_count[0], 1ull, __ATOMIC_RELAXED);
		return cr_check_hip(launch_generic(plan, &g, in_valid_bytes, stream), "k_generic launch");
	}
}

/* Variable rate in ONE launch (k_generic_segments): uploads the table of `count` non-empty segments behind whatever is queued on
   `stream` and enqueues the kernel there; nothing is waited for except - rarely - the previous call's use of the table buffers. */
int cr_segments_run(const ClownResamplerAMD_Plan *plan, const void *d_in, void *d_out, const crhip_segment *segments, size_t count,
                    uint64_t n_out, int out_s16, void *stream)
{
	cr_device_ctx *ctx = ensure_ctx(plan->device);
	crhip_segments_launch l;
	int bad = 0;

	if (ctx == NULL)
		return -1;
	if (count == 0 || n_out == 0)
		return 0;

	pthread_mutex_lock(&ctx->workspace_lock);
	if (ctx->seg_in_use && cr_check_hip(crhip_event_sync(ctx->seg_event), "hipEventSynchronize") != 0)
		bad = 1;
	ctx->seg_in_use = 0;
	if (!bad && ctx->seg_capacity < count)
	{
		const size_t want = count + count / 4 + 64;

		if (ctx->seg_host != NULL)
			host_free(ctx->seg_host);
		dev_free(ctx->seg_dev);
		ctx->seg_host = NULL;
		ctx->seg_dev = NULL;
		ctx->seg_capacity = 0;
		if (cr_check_hip(host_alloc((void **)&ctx->seg_host, want * sizeof(crhip_segment)), "hipHostMalloc(segment table)") != 0
		 || cr_check_hip(dev_malloc((void **)&ctx->seg_dev, want * sizeof(crhip_segment)), "hipMalloc(segment table)") != 0)
			bad = 1;
		else
			ctx->seg_capacity = want;
	}
	if (!bad && ctx->seg_event == NULL && cr_check_hip(crhip_event_create(&ctx->seg_event), "hipEventCreate") != 0)
		bad = 1;

	if (!bad)
	{
		memcpy(ctx->seg_host, segments, count * sizeof(crhip_segment));
		memset(&l, 0, sizeof(l));
		l.d_in = d_in;
		l.d_out = d_out;
		l.d_table = plan->d_table;
		l.d_segments = ctx->seg_dev;
		l.n_out = n_out;
		l.n_segments = (uint32_t)count;
		l.table_len = plan->table_len;
		l.channels = plan->channels;
		l.out_s16 = out_s16 ? 1u : 0u;
		bad = cr_check_hip(crhip_memcpy_h2d(ctx->seg_dev, ctx->seg_host, count * sizeof(crhip_segment), stream), "hipMemcpyAsync(segment table)") != 0
		   || cr_check_hip(launch_segments(plan, &l, stream), "k_generic_segments launch") != 0
		   || cr_check_hip(crhip_event_record(ctx->seg_event, stream), "hipEventRecord") != 0;
		ctx->seg_in_use = 1;   /* (also after a failure part-way: the copy may be queued) */
		__atomic_fetch_add(&g_launch_count[0], 1ull, __ATOMIC_RELAXED);
	}
	pthread_mutex_unlock(&ctx->workspace_lock);
	return bad ? -1 : 0;
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
		dev_free(*p);
	*p = NULL;
	*have = 0;

	if (cr_check_hip(dev_malloc((void **)p, want), "hipMalloc(staging)") != 0)
		return -1;

	*have = want;
	return 0;
}

static cr_workspace *workspace_acquire(cr_device_ctx *ctx, size_t in_bytes, size_t out_bytes)
{
	pthread_mutex_lock(&ctx->workspace_lock);

	if (ctx->workspace.stream == NULL && cr_check_hip(crhip_stream_create(&ctx->workspace.stream), "hipStreamCreate") != 0)
		goto fail;

	if (grow(&ctx->workspace.d_in, &ctx->workspace.d_in_bytes, in_bytes + 64) != 0 || grow(&ctx->workspace.d_out, &ctx->workspace.d_out_bytes, out_bytes + 64) != 0)
		goto fail;

	return &ctx->workspace;

fail:
	pthread_mutex_unlock(&ctx->workspace_lock);
	return NULL;
}

static void workspace_release(cr_device_ctx *ctx)
{
	pthread_mutex_unlock(&ctx->workspace_lock);
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
	int device;
	int hip_error; /* first failing HIP call of the thread (reported by the caller's thread, whose error state the API exposes) */
} cr_download;

static void *download_thread(void *arg)
{
	cr_download *d = (cr_download *)arg;
	int code = crhip_set_device(d->device);

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
			code = copy_pageable_d2h(d->slot[s].host_dst, d->slot[s].dev_src, d->slot[s].bytes, d->stream);
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
	const int pipelined = n_out > batch_frames && !g_env.no_host_pipeline;
	uint64_t done = 0, batch = 0;
	cr_download dl;
	pthread_t thread;
	int have_thread = 0, bad = 0;
	cr_workspace *ws;
	cr_device_ctx *ctx;
	void *alias_in = NULL, *alias_out = NULL;

	if (n_out == 0)
		return 0;

	ctx = ensure_ctx(plan->device);
	if (ctx == NULL)
		return -1;

	/* SMALL calls (a sound-card sized request, a refill of the streaming API): two hipMemcpyAsync and a launch cost ~35 us
	   whatever the size, most of it in the two copies.  Up to CR_SMALL_CALL_BYTES the kernel therefore works on pinned host
	   memory directly - the input is copied into it by the CPU, the LDS-DMA reads it and the stores write the result across
	   PCIe, and one synchronise later the CPU copies the result out: one launch, no copy calls (480 frames: 34 -> ~15 us). */
	if (n_out <= batch_frames && !g_env.no_small_call_path)
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
			ws = workspace_acquire(ctx, 0, 0); /* (the lock and the stream) */
			if (ws == NULL)
				return -1;
			if (ctx->small == NULL && cr_check_hip(host_alloc((void **)&ctx->small, CR_SMALL_CALL_BYTES), "hipHostMalloc(small-call block)") != 0)
			{
				ctx->small = NULL;
				workspace_release(ctx);
				return -1;
			}
			memcpy(ctx->small, host_in + pi * plan->channels, in_bytes);
			bad = cr_plan_launch(plan, ctx->small, in_bytes, ctx->small + out_at, 0, pos_frac, n_out, ws->stream, out_s16) != 0
			   || cr_check_hip(crhip_stream_sync(ws->stream), "hipStreamSynchronize") != 0;
			if (!bad)
				memcpy(host_out, ctx->small + out_at, out_bytes);
			workspace_release(ctx);
			return bad ? -1 : 0;
		}
	}

	/* PAGE-LOCKED caller buffers (hipHostMalloc / hipHostRegister - a client that did the right thing): the device can address them,
	   so nothing needs staging.  host_direct 2: ONE launch over the whole call, the LDS-DMA reading the caller's input across the bus
	   and the stores writing the caller's output, both directions of the link busy at once, no copy calls, no helper thread.
	   host_direct 1: only the input is read in place (no uploads); the output still goes through device staging and the download
	   thread.  Which of the three a call takes is a measured rule (profiles/r04_host_paths_pinned.log); pageable memory always stages. */
	if (g_env.host_direct != 0)
	{
		/* (everything from the call's first frame to the end of the caller's padded buffer: what the staged path may read as well) */
		const uint64_t readable = pos_int < in_frames ? in_frames - pos_int : 0;

		if (readable != 0 && crhip_host_alias(host_in + pos_int * plan->channels, (size_t)readable * frame_in, &alias_in) != 0)
			alias_in = NULL;
		if (alias_in != NULL && (g_env.host_direct == 2 || g_env.host_direct < 0)
		 && crhip_host_alias(host_out, (size_t)n_out * frame_out, &alias_out) != 0)
			alias_out = NULL;
		if (g_env.host_direct < 0 && alias_out == NULL)
			alias_in = NULL;   /* (the rule: all or nothing) */

		if (alias_in != NULL && alias_out != NULL)
		{
			ws = workspace_acquire(ctx, 0, 0); /* (the lock and the stream) */
			if (ws == NULL)
				return -1;
			bad = cr_plan_launch(plan, alias_in, readable * frame_in, alias_out, 0, pos_frac, n_out, ws->stream, out_s16) != 0
			   || cr_check_hip(crhip_stream_sync(ws->stream), "hipStreamSynchronize") != 0;
			workspace_release(ctx);
			return bad ? -1 : 0;
		}
	}

	/* both staging sets, sized for a full batch, under the one workspace lock for the whole call */
	{
		const uint64_t n0 = n_out < batch_frames ? n_out : batch_frames;
		uint64_t extent0 = cr_input_extent(&plan->cfg, 0, 65535u, plan->increment, n0);

		if (extent0 > in_frames)
			extent0 = in_frames;
		ws = workspace_acquire(ctx, (size_t)extent0 * frame_in, (size_t)n0 * frame_out);
		if (ws == NULL)
			return -1;

		if (pipelined)
		{
			int k, failed = ctx->workspace_more[0].stream == NULL && cr_check_hip(crhip_stream_create(&ctx->workspace_more[0].stream), "hipStreamCreate") != 0;

			for (k = 0; k < CR_EXTRA_SETS && !failed; ++k)
				failed = grow(&ctx->workspace_more[k].d_in, &ctx->workspace_more[k].d_in_bytes, (size_t)extent0 * frame_in + 64) != 0
				      || grow(&ctx->workspace_more[k].d_out, &ctx->workspace_more[k].d_out_bytes, (size_t)n0 * frame_out + 64) != 0;
			if (failed)
			{
				workspace_release(ctx);
				return -1;
			}

			memset(&dl, 0, sizeof(dl));
			dl.device = plan->device;
			pthread_mutex_init(&dl.lock, NULL);
			pthread_cond_init(&dl.changed, NULL);
			/* uploads and kernels all go to ONE stream (ws->stream), downloads all to another (the second set's): copies
			   of one direction per stream is what lets the runtime run the two directions side by side (measured: with
			   each batch's three steps on its own stream the download stalled for as long as the next upload ran) */
			dl.stream = ctx->workspace_more[0].stream;
			for (k = 0; k < 1 + CR_EXTRA_SETS && !failed; ++k)
				failed = cr_check_hip(crhip_event_create(&dl.slot[k].ready), "hipEventCreate") != 0;
			if (failed)
			{
				for (k = 0; k < 1 + CR_EXTRA_SETS; ++k)
					if (dl.slot[k].ready != NULL)
						crhip_event_destroy(dl.slot[k].ready);
				workspace_release(ctx);
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
		cr_workspace *w = set == 0u ? ws : &ctx->workspace_more[set - 1u];

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

		if (alias_in != NULL)   /* the input is read where it lies (alias_in is the device's address of host_in + pos_int frames) */
			bad = cr_plan_launch(plan, (const unsigned char *)alias_in + (size_t)(pi - pos_int) * frame_in, extent * frame_in, w->d_out, 0, pf, n, ws->stream, out_s16) != 0;
		else
			bad = cr_check_hip(copy_pageable_h2d(w->d_in, host_in + pi * plan->channels, (size_t)extent * frame_in, ws->stream), "hipMemcpyAsync(H2D)") != 0
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
			bad = cr_check_hip(copy_pageable_d2h((unsigned char *)host_out + done * frame_out, w->d_out, (size_t)n * frame_out, ws->stream), "hipMemcpyAsync(D2H)") != 0
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
			crhip_stream_sync(ctx->workspace_more[0].stream);
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

	workspace_release(ctx);
	return bad ? -1 : 0;
}

int cr_run_single_frame(const ClownResamplerAMD_Plan *plan, const int16_t *host_window, uint64_t window_frames,
                        uint64_t pos_frac, const int64_t *acc_in, int64_t *acc_out)
{
	const size_t in_bytes = (size_t)window_frames * plan->channels * sizeof(int16_t);
	const size_t acc_bytes = (size_t)plan->channels * sizeof(int64_t);
	cr_device_ctx *ctx = ensure_ctx(plan->device);
	cr_workspace *ws = ctx != NULL ? workspace_acquire(ctx, in_bytes, 2 * acc_bytes) : NULL;
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

	bad = cr_check_hip(copy_pageable_h2d(ws->d_in, host_window, in_bytes, ws->stream), "hipMemcpyAsync(H2D)") != 0
	   || cr_check_hip(crhip_memcpy_h2d(ws->d_out + acc_bytes, acc_in, acc_bytes, ws->stream), "hipMemcpyAsync(H2D)") != 0
	   || cr_check_hip(launch_generic(plan, &g, in_bytes, ws->stream), "k_generic launch") != 0
	   || cr_check_hip(crhip_memcpy_d2h(acc_out, ws->d_out, acc_bytes, ws->stream), "hipMemcpyAsync(D2H)") != 0
	   || cr_check_hip(crhip_stream_sync(ws->stream), "hipStreamSynchronize") != 0;

	workspace_release(ctx);
	return bad ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------------------- */
/* plan introspection and debug switches (public, radius-independent)                                                         */
/* ------------------------------------------------------------------------------------------------------- */

void ClownResamplerAMD_PlanGetInfo(const ClownResamplerAMD_Plan *plan, ClownResamplerAMD_PlanInfo *info)
{
	memset(info, 0, sizeof(*info));
	info->kernel = plan->use_poly ? (plan->variant == CRHIP_VARIANT_RT_WAVE2S ? 6u : plan->vecs >= 200u ? 3u : plan->vecs >= 150u ? 4u : plan->vecs >= 100u ? 2u : 1u) : 0u;
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
	if (plan->use_poly && plan->brief.below != 0)
	{
		info->brief_kernel = plan->brief.vecs >= 150u ? 4u : plan->brief.vecs >= 100u ? 2u : 1u;
		info->brief_variant = plan->brief.variant;
		info->brief_below = plan->brief.below;
	}
}

int cr_env_no_replay_thread(void)
{
	env_ready();
	return g_env.no_replay_thread;
}

static int g_segments_mode = 0;

int cr_segments_mode(void)
{
	return g_segments_mode;
}

void ClownResamplerAMD_DebugSegmentsMode(int mode)
{
	g_segments_mode = (mode >= 0 && mode <= 2) ? mode : 0;
}

/* 1 when the host-pointer entry points would read / write [host, host + bytes) IN PLACE (page-locked memory the current device can address:
   hipHostMalloc, hipHostRegister), 0 when they stage it (pageable memory) - the question cr_run_host asks of every larger call. */
int ClownResamplerAMD_DebugHostIsDeviceVisible(const void *host, size_t bytes)
{
	void *alias = NULL;

	if (cr_ensure_device() != 0)
		return 0;
	return crhip_host_alias(host, bytes, &alias) == 0 && alias != NULL;
}

/* The library at REST (no call in progress on any thread): what its process-wide state must look like then.  Waits for the devices it has
   touched, so every launch it ever enqueued has finished.  0 and an empty message when everything holds; otherwise the number of
   findings, the first few in `message`.  (the reference has no globals; this library has - plan cache, ticket rings,
   capture pool, staging sets, test hooks - and the GPU tests check after every test that they have returned to rest.) */
int ClownResamplerAMD_DebugSelfCheck(char *message, size_t capacity)
{
	int findings = 0, d;
	size_t at = 0;
	const ClownResamplerAMD_Plan *plan;
	const int previous_device = current_device();

#define FINDING(...) \
	do \
	{ \
		++findings; \
		if (message != NULL && at + 1 < capacity) \
		{ \
			const int n_ = snprintf(message + at, capacity - at, __VA_ARGS__); \
			if (n_ > 0) \
				at += (size_t)n_ < capacity - at ? (size_t)n_ : capacity - at - 1; \
			if (at + 2 < capacity) \
			{ \
				message[at++] = ';'; \
				message[at++] = ' '; \
				message[at] = '\0'; \
			} \
		} \
	} while (0)

	if (message != NULL && capacity != 0)
		message[0] = '\0';

	/* test hooks back at their defaults */
	if (g_force_generic) FINDING("DebugForceGenericKernel is still on");
	if (g_no_int_kernel) FINDING("DebugDisableIntKernel is still on");
	if (g_no_dual_mono) FINDING("DebugDisableDualMono is still on");
	if (g_seg_mode != 0) FINDING("DebugSegKernel is still %d", g_seg_mode);
	if (g_segments_mode != 0) FINDING("DebugSegmentsMode is still %d", g_segments_mode);
	if (g_variant >= 0 && g_variant != CR_DEFAULT_VARIANT) FINDING("DebugSetVariant is still %d", g_variant);
	if (g_debug_stamps != NULL) FINDING("DebugSetStampBuffer is still set");

	pthread_mutex_lock(&g_lock);

	/* plans: nobody holds one; a store's reference count = the plans (and dual-mono partners) that view it */
	for (plan = g_plans; plan != NULL; plan = plan->next)
	{
		const ClownResamplerAMD_Plan *other;
		int views = 0;

		if (plan->users != 0)
			FINDING("plan %p (%u ch, increment %llu) is held by %u calls", (const void *)plan, plan->channels, (unsigned long long)plan->increment, plan->users);
		if (plan->store == NULL)
		{
			FINDING("plan %p has no store", (const void *)plan);
			continue;
		}
		for (other = g_plans; other != NULL; other = other->next)
		{
			views += other->store == plan->store;
			views += other->dual.partner != NULL && other->dual.partner->store == plan->store;
		}
		if (views != plan->store->refs)
			FINDING("store %p: %d references counted, %d plans view it", (const void *)plan->store, plan->store->refs, views);
		if (plan->d_table != plan->store->d_table || (plan->use_poly && plan->d_rows != plan->store->d_rows) || (plan->seg.available && plan->seg.d_rows != plan->store->d_rows_seg))
			FINDING("plan %p: its views of table / rows are not its store's", (const void *)plan);
		if (plan->dual.partner != NULL && (plan->dual.partner->store != plan->store || plan->dual.partner->d_rows != plan->store->d_rows))
			FINDING("plan %p: the dual-mono partner views other rows", (const void *)plan);
	}

	for (d = 0; d < CR_MAX_DEVICES; ++d)
	{
		cr_device_ctx *ctx = g_ctx[d];
		unsigned r;
		uint32_t *host;
		const size_t ring_words = (size_t)CR_RING_SLOTS * CRHIP_TICKET_WORDS;

		if (ctx == NULL || !ctx->ready)
			continue;
		if (crhip_set_device(d) != 0 || crhip_device_sync() != 0)
		{
			FINDING("device %d: hipDeviceSynchronize failed", d);
			continue;
		}
		if (pthread_mutex_trylock(&ctx->workspace_lock) != 0)
			FINDING("device %d: the staging workspace is locked", d);
		else
		{
			if (ctx->seg_in_use && ctx->seg_event != NULL && crhip_event_sync(ctx->seg_event) != 0)
				FINDING("device %d: the segment table's event does not complete", d);
			pthread_mutex_unlock(&ctx->workspace_lock);
		}
		if (pthread_mutex_trylock(&ctx->ring_lock) != 0)
		{
			FINDING("device %d: the ticket rings are locked", d);
			continue;
		}
		/* every ticket block of every ring reads zero: each launch leaves its block as it found it */
		host = (uint32_t *)malloc(ring_words * sizeof(uint32_t));
		for (r = 0; host != NULL && r < ctx->ring_count; ++r)
		{
			size_t w;

			if (ctx->rings[r].pending != 0)
				FINDING("device %d ring %u: %u launches drawn and not enqueued", d, r, ctx->rings[r].pending);
			if (copy_pageable_d2h(host, ctx->rings[r].blocks, ring_words * sizeof(uint32_t), NULL) != 0 || crhip_stream_sync(NULL) != 0)
			{
				FINDING("device %d ring %u: cannot be read back", d, r);
				continue;
			}
			for (w = 0; w < ring_words; ++w)
				if (host[w] != 0)
				{
					FINDING("device %d ring %u: block %lu word %lu reads %u, not 0", d, r, (unsigned long)(w / CRHIP_TICKET_WORDS), (unsigned long)(w % CRHIP_TICKET_WORDS), host[w]);
					break;
				}
		}
		/* the part of the capture pool that has not been handed out reads zero too */
		if (host != NULL && ctx->capture_left != 0)
		{
			const size_t words = ctx->capture_left * CRHIP_TICKET_WORDS < ring_words ? ctx->capture_left * CRHIP_TICKET_WORDS : ring_words;
			size_t w;

			if (copy_pageable_d2h(host, ctx->capture_at, words * sizeof(uint32_t), NULL) == 0 && crhip_stream_sync(NULL) == 0)
				for (w = 0; w < words; ++w)
					if (host[w] != 0)
					{
						FINDING("device %d: unused capture block word %lu reads %u, not 0", d, (unsigned long)w, host[w]);
						break;
					}
		}
		free(host);
		pthread_mutex_unlock(&ctx->ring_lock);
	}
	pthread_mutex_unlock(&g_lock);
	if (previous_device >= 0 && g_ctx[previous_device] != NULL && g_ctx[previous_device]->ready)
		crhip_set_device(previous_device);
#undef FINDING
	return findings;
}

uint32_t ClownResamplerAMD_PlanDualMonoKernel(const ClownResamplerAMD_Plan *plan)
{
	const ClownResamplerAMD_Plan *partner = plan != NULL ? plan->dual.partner : NULL;

	if (partner == NULL || g_no_dual_mono)
		return 0u;
	return partner->vecs >= 150u ? 4u : 1u;
}

uint32_t ClownResamplerAMD_PlanSegKernel(const ClownResamplerAMD_Plan *plan)
{
	return (plan != NULL && plan->seg.available && g_seg_mode != 2) ? (uint32_t)CR_COUNT_SEG : 0u;
}

uint32_t ClownResamplerAMD_PlanPaddedTiles(const ClownResamplerAMD_Plan *plan)
{
	return plan != NULL && plan->use_poly ? plan->padded : 0u;
}

void ClownResamplerAMD_DebugSegKernel(int mode)
{
	g_seg_mode = mode;
}

void ClownResamplerAMD_DebugDisableDualMono(int on)
{
	g_no_dual_mono = on != 0;
}

void ClownResamplerAMD_DebugDisableIntKernel(int on)
{
	g_no_int_kernel = on != 0;
}

unsigned long long ClownResamplerAMD_DebugLaunchCount(unsigned kernel)
{
	return kernel < CR_KERNEL_IDS ? __atomic_load_n(&g_launch_count[kernel], __ATOMIC_RELAXED) : 0ull;
}

uint32_t ClownResamplerAMD_PlanKernelAt(const ClownResamplerAMD_Plan *plan, uint32_t position_fractional)
{
	ClownResamplerAMD_PlanInfo info;
	crhip_int_launch il;
	uint32_t first_slot;

	if (plan->use_poly && !g_force_generic && !g_no_int_kernel && plan->intk.available && int_launch_row(plan, position_fractional & 0xFFFFu, &il, &first_slot))
		return 5u;
	ClownResamplerAMD_PlanGetInfo(plan, &info);
	return g_force_generic ? 0u : info.kernel;
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
	g_variant = ((variant >= 0 && variant < crhip_poly_variants()) || diagnostic_variant(variant)) ? variant : CR_DEFAULT_VARIANT;
}

void ClownResamplerAMD_DebugForceGenericKernel(int on)
{
	g_force_generic = on != 0;
}

/* ------------------------------------------------------------------------------------------------------- */
/* streaming side windows                                                                                  */
/* ------------------------------------------------------------------------------------------------------- */

/* The side window of the high-level state at address `owner`: the one already registered for that address (the state is being
   re-initialised, or a new state lives where a discarded one did - either way the old window's contents are dead), else a new
   one.  Nothing of the caller's (possibly uninitialised) state is read to find it. */
cr_stream *cr_stream_claim(const void *owner)
{
	cr_stream *stream;

	pthread_mutex_lock(&g_lock);
	for (stream = g_streams; stream != NULL; stream = stream->next)
		if (stream->owner == owner)
			break;
	if (stream == NULL)
	{
		stream = (cr_stream *)calloc(1, sizeof(*stream));
		if (stream != NULL)
		{
			stream->owner = owner;
			stream->next = g_streams;
			g_streams = stream;
		}
	}
	if (stream != NULL)
	{
		stream->id = ++g_stream_serial ^ 0x434C4F574E5253ull; /* never 0; a new id per claim: keys of the address's past lives die */
		stream->pull_count = 0;
		/* an idle window does not keep its largest size for ever */
		if (stream->window_samples > (size_t)1 << 16)
		{
			free(stream->window);
			stream->window = NULL;
			stream->window_samples = 0;
		}
	}
	pthread_mutex_unlock(&g_lock);
	return stream;
}

/* the window of the state at `owner` whose key says `id`; NULL when the key is stale or the bytes were copied elsewhere */
cr_stream *cr_stream_lookup(uint64_t id, const void *owner)
{
	cr_stream *stream;

	pthread_mutex_lock(&g_lock);
	for (stream = g_streams; stream != NULL; stream = stream->next)
		if (stream->id == id && stream->owner == owner)
			break;
	pthread_mutex_unlock(&g_lock);
	return stream;
}

size_t ClownResamplerAMD_StreamingWindowCount(void)
{
	const cr_stream *stream;
	size_t n = 0;

	pthread_mutex_lock(&g_lock);
	for (stream = g_streams; stream != NULL; stream = stream->next)
		++n;
	pthread_mutex_unlock(&g_lock);
	return n;
}

/* frees the window registered for `owner`, if any */
void cr_stream_drop(const void *owner)
{
	cr_stream **link, *stream = NULL;

	pthread_mutex_lock(&g_lock);
	for (link = &g_streams; *link != NULL; link = &(*link)->next)
		if ((*link)->owner == owner)
		{
			stream = *link;
			*link = stream->next;
			break;
		}
	pthread_mutex_unlock(&g_lock);
	if (stream != NULL)
	{
		free(stream->window);
		free(stream->pull_ends);
		free(stream);
	}
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

int cr_stream_note_pull(cr_stream *stream, size_t end_sample)
{
	if (stream->pull_count == stream->pull_capacity)
	{
		const size_t capacity = stream->pull_capacity != 0 ? 2 * stream->pull_capacity : 64;
		size_t *grown = (size_t *)realloc(stream->pull_ends, capacity * sizeof(size_t));

		if (grown == NULL)
			return -1;

		stream->pull_ends = grown;
		stream->pull_capacity = capacity;
	}

	stream->pull_ends[stream->pull_count++] = end_sample;
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
