/*
 * cr_context.h - process-wide GPU context of libclownresampler_amd.so: error reporting, device selection,
 * the plan cache and the staging workspace used by the host-buffer entry points.  Radius-independent; the
 * per-radius API instances (cr_api.c) sit on top.  Internal.
 */
#ifndef CR_CONTEXT_H
#define CR_CONTEXT_H

#include <stddef.h>
#include <stdint.h>

#include "cr_plan.h"
#include "crhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Reports through the installed error handler (default: print + abort).  Returns `code`. */
int cr_fail(int code, const char *format, ...);
/* Number of cr_fail calls made on this thread so far. */
unsigned long cr_error_serial(void);
/* For threads the library itself creates: with `on`, cr_fail on THIS thread only records (no handler call, no abort); the creating
   thread then collects code and message with cr_error_take on the worker's behalf... */
void cr_error_defer(int on);
/* ... called ON the worker thread before it ends: returns the recorded code (0: none), copies the message, clears the record. */
int cr_error_take(char *message, size_t capacity);
/* cr_fail(CLOWNRESAMPLER_AMD_ERROR_HIP, ...) when hip_code != 0; returns hip_code. */
int cr_check_hip(int hip_code, const char *what);

/* Makes the calling thread's device (ClownResamplerAMD_SetThreadDevice, else the process default) current for HIP and makes
   sure its context exists.  0 on success. */
int cr_ensure_device(void);
int cr_current_device(void);
const crhip_device_info *cr_device_info(void);

/* What plans that differ only in their increment share: the polyphase rows are a function of the fractional position,
   the configuration and the table - not of the increment, which only decides which fractions a stream visits.  A
   variable-rate client that keeps its low-pass (every pure upsampler: clownresampler.h:968 gives the same
   configuration for any output rate >= the input rate) therefore pays for the rows once.  Reference-counted. */
typedef struct cr_plan_store
{
	int refs;
	uint32_t table_len;
	int32_t *d_table;       /* caller's table as int32 (generic kernel) */
	cr_poly poly;           /* host copy of the rows + row-index form (owns its arrays) */
	int32_t *d_rows;        /* device image of the rows, NULL until a plan needs it */
	void *d_rows_seg;       /* k_seg's image of the same rows (64 bytes per row: 15 x |weight| / 65536 as float, 2 * reciprocal), NULL until a plan needs it */
	int rows_layout;        /* CR_IMAGE_* of d_rows */
	uint32_t plane_rows, swizzle, device_row_stride;
} cr_plan_store;

/* The plan object behind the public opaque ClownResamplerAMD_Plan. */
typedef struct ClownResamplerAMD_Plan
{
	struct ClownResamplerAMD_Plan *next;
	/* key */
	uint64_t table_hash;
	unsigned radius;
	cr_config cfg;
	uint32_t channels;
	uint64_t increment;
	int device;
	uint32_t key_variant;   /* the variant that was current when the plan was made (`variant` below is the one it ended up with) */
	/* cache bookkeeping (under the context lock) */
	int pinned;             /* handed out by ClownResamplerAMD_PlanCreate: never evicted */
	unsigned users;         /* calls in progress that hold the plan (cr_plan_get .. cr_plan_release) */
	uint64_t last_use;
	/* contents: `poly`, `d_table` and `d_rows` are VIEWS of the store's */
	cr_plan_store *store;
	uint32_t table_len;
	int32_t *d_table;
	cr_poly poly;
	int use_poly;
	const char *generic_reason;
	int32_t *d_rows;
	uint32_t threads, vecs, tile_frames, lds_bytes, max_blocks, specialised, variant;
	uint32_t max_blocks_s16; /* persistent-grid cap of the int16-output form (its own function, its own register footprint) */
	uint32_t plane_rows, swizzle;
	uint32_t wave_tile;     /* k_wave2s: output frames per wave-tile */
	uint32_t lane_map;      /* k_wave2: which frame of its 64 a lane takes (crhip_poly_launch.lane_map), chosen for this plan's increment */
	uint32_t lds_swizzle;   /* k_wave2: the rotation it applies while staging the (plain) rows into LDS, chosen for this plan's increment */
	uint32_t device_row_stride; /* int32 per row of the device image (COMPACT for specialised instances, SPLIT otherwise) */
	double conflict_plain, conflict_best; /* modelled extra LDS cycles per row read without / with the swizzle */
	/* k_up's wave-tiles are long (64 input positions: 757 frames at 12x) and a wave computes one in ~15 us however few there
	   are: launches that leave a wave fewer than a handful go to the instance's other kernel (k_wave2 / k_wave / k_poly, the
	   one k_up falls back to) over the SAME rows image - `brief` is that kernel's shape, for launches of fewer than `below`
	   output frames (0: none) */
	struct
	{
		uint64_t below;
		uint32_t threads, vecs, tile_frames, lds_bytes, max_blocks, max_blocks_s16, variant, lds_swizzle, lane_map;
	} brief;
	/* whole-number downsampling ratios (increment = ratio << 16) and ratios that repeat after 2 or 4 frames (increment * period =
	   ratio << 16: 3:2, 1:2, 1:4): k_int (cr_kint.hpp) where there is an instance for (channels, ratio : period, slots).  Whether a LAUNCH takes it depends on its fractional position - the row that fraction selects
	   must have the instance's slot signs (cr_plan_launch checks; a stream that starts from Init stays at fraction 0). */
	struct
	{
		int available;
		uint32_t ratio, period;   /* `ratio` input frames per `period` output frames (period 1: the whole-number ratios; 2, 4: 3:2, 1:2, 1:4 ...) */
		crhip_int_shape shape;
		uint32_t max_blocks, max_blocks_s16;
	} intk;
	/* k_seg (cr_kseg.hpp): long launches of a k_up2 shape (stereo, 15 slots, fixed slot signs) with the lanes of a wave on frames of equal
	   fraction, `period` = 65536 / gcd(increment, 65536) output frames apart (or a multiple), the row in scalar registers */
	struct
	{
		int available;
		uint64_t period;
		uint32_t threads, lds_bytes, max_blocks, chunk;
		const void *d_rows;
	} seg;
	uint32_t padded;   /* 1: k_poly's run-time-slot instance computes from padded tiles (crhip_poly_launch.padded) */
	/* DUAL MONO (mono plans; crhip_poly_launch.dual, cr_kpoly.hpp): long launches run on the STEREO instance of the same
	   configuration, output frames j and j + H as its two channels (H * increment a multiple of 65536: equal fractions, one row for
	   both).  `partner` is a private stereo plan over this plan's own rows (not in the cache; freed with this plan); `period` =
	   65536 / gcd(increment, 65536): H must be a multiple of it. */
	struct
	{
		struct ClownResamplerAMD_Plan *partner;
		uint64_t period;
		uint32_t max_blocks;
	} dual;
} ClownResamplerAMD_Plan;

/* Cache lookup by (hash of the caller's raw table bytes, radius, configuration, channels, increment); on a miss
   `fill_table(user, dst)` is asked to write the table as int32 and the plan is built (sharing the rows of a plan that
   differs only in its increment, if there is one) and uploaded.  NULL after cr_fail.
   The caller holds the plan until cr_plan_release; with `pin` the plan additionally stays valid until Shutdown.
   Unpinned plans nobody holds are dropped, least recently used first, once there are more than
   ClownResamplerAMD_SetPlanCacheLimit of them. */
typedef int (*cr_table_fill)(const void *user, int32_t *dst, size_t count);
ClownResamplerAMD_Plan *cr_plan_get(uint64_t table_hash, size_t table_len, cr_table_fill fill_table, const void *user,
                                    unsigned radius, const cr_config *cfg, uint32_t channels, uint64_t increment, int pin);
void cr_plan_release(const ClownResamplerAMD_Plan *plan);
/* ... and the same on a given device (the multi-device entry point): the calling thread's HIP device is `device` afterwards */
ClownResamplerAMD_Plan *cr_plan_get_on(int device, uint64_t table_hash, size_t table_len, cr_table_fill fill_table, const void *user,
                                       unsigned radius, const cr_config *cfg, uint32_t channels, uint64_t increment, int pin);
/* Makes the plan's device current for HIP on the calling thread.  0 on success. */
int cr_ensure_device_of(const ClownResamplerAMD_Plan *plan);

/* Enqueues the computation of output frames [0, n_out) starting at (pos_int, pos_frac) on `stream`. 0 on success. */
int cr_plan_launch(const ClownResamplerAMD_Plan *plan, const void *d_in, uint64_t in_valid_bytes, void *d_out,
                   uint64_t pos_int, uint64_t pos_frac, uint64_t n_out, void *stream, int out_s16);

/* the 32-bit descriptor arithmetic of the dual-mono kernels holds for this launch (cr_context.c) */
int cr_dual_mono_fits(uint64_t n_out, uint64_t half, uint64_t tile_frames, uint64_t increment, uint64_t in_valid_bytes);

/* Variable rate in one launch: `count` non-empty segments (first_out ascending) of one timeline at d_in, n_out frames in all, through
   the generic kernel with a segment table (any configuration per segment; `plan` supplies the device, the table and the channel
   count).  Enqueued on `stream`, not waited for.  0 on success. */
int cr_segments_run(const ClownResamplerAMD_Plan *plan, const void *d_in, void *d_out, const crhip_segment *segments, size_t count,
                    uint64_t n_out, int out_s16, void *stream);
/* 0: the measured rule decides between one launch per segment and one launch for all, 1: always per segment, 2: always one launch */
int cr_segments_mode(void);
/* CLOWNRESAMPLER_AMD_NO_REPLAY_THREAD is set: ClownResampler_LowLevel_Resample stays on the calling thread for the whole call */
int cr_env_no_replay_thread(void);

/* cr_multi.c: the multi-device call behind ClownResamplerAMD_ResampleShardedDevice, and its part of Shutdown */
struct ClownResampler_LowLevel_State;
struct ClownResamplerAMD_DeviceShard;
size_t cr_resample_sharded(struct ClownResampler_LowLevel_State *resampler, uint64_t table_hash, size_t table_len, cr_table_fill fill_table, const void *table_user,
                           unsigned radius, size_t total_input_frames, const struct ClownResamplerAMD_DeviceShard *shards, unsigned shard_count,
                           int output_is_s16, int gather_mode, unsigned root_shard, void *root_output);
void cr_multi_shutdown(void);

/* Staging workspace for the host-buffer entry points: one per device, handed out under that device's lock. */
typedef struct cr_workspace
{
	void *stream;
	unsigned char *d_in;
	size_t d_in_bytes;
	unsigned char *d_out;
	size_t d_out_bytes;
} cr_workspace;

/* Computes output frames [0, n_out) from HOST input into a HOST int32 buffer: uploads the input window, launches,
   downloads, synchronises.  `host_in` points at padded-buffer frame 0 and in_frames frames are readable. 0 on success. */
int cr_run_host(const ClownResamplerAMD_Plan *plan, const int16_t *host_in, uint64_t in_frames, uint64_t pos_int,
                uint64_t pos_frac, uint64_t n_out, void *host_out, int out_s16);

/* One frame with incoming accumulators, 64-bit results (ClownResampler_LowestLevel_Resample). 0 on success. */
int cr_run_single_frame(const ClownResamplerAMD_Plan *plan, const int16_t *host_window, uint64_t window_frames,
                        uint64_t pos_frac, const int64_t *acc_in, int64_t *acc_out);

/* Side window of one high-level (streaming) state: the reference's fixed 0x1000-sample staging buffer
   (clownresampler.h:654, with the TODO there) made large, so that a whole batch of input pulls becomes one GPU call. */
typedef struct cr_stream
{
	uint64_t id;
	const void *owner;      /* address of the ClownResampler_HighLevel_State the window belongs to */
	struct cr_stream *next;
	int16_t *window;        /* [left halo | unconsumed frames ... | look-ahead], same layout as the reference's buffer */
	size_t window_samples;  /* allocated */
	size_t start, end;      /* sample indices into window: the frames not yet resampled (input_buffer_start / _end) */
	size_t target_frames;   /* payload frames to collect per refill; grows while the consumer keeps draining whole windows */
	/* where each of the refill's pulls ended (sample indices into window, ascending, the last one == end): the windows the
	   REFERENCE would have resampled one by one - what its state says after a consumer's stop depends on them (cr_api.c) */
	size_t *pull_ends;
	size_t pull_count, pull_capacity;
} cr_stream;

cr_stream *cr_stream_claim(const void *owner);  /* the window of the state at this address (new id; made if there is none) */
cr_stream *cr_stream_lookup(uint64_t id, const void *owner);
void cr_stream_drop(const void *owner);
int cr_stream_reserve(cr_stream *stream, size_t samples);   /* grows window keeping its contents; 0 on success */
int cr_stream_note_pull(cr_stream *stream, size_t end_sample);   /* appends to pull_ends; 0 on success */
size_t cr_stream_max_frames(void);             /* ClownResamplerAMD_SetStreamingWindow */

#ifdef __cplusplus
}
#endif

#endif /* CR_CONTEXT_H */
