/*
 * crhip.h - the thin C-ABI between the host code (C: cr_plan.c, cr_context.c, cr_api.c) and the HIP
 * translation unit (cr_kernels.hip).  Plain pointers, sizes and fixed-width integers only.
 * Internal to libclownresampler_amd.so; the public surface is include/clownresampler*.h.
 */
#ifndef CRHIP_H
#define CRHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRHIP_MAX_CHANNELS 16

/* Row-index formulas (device and host mirror must agree; see cr_plan.c). */
#define CRHIP_ROWMODE_AFFINE 0   /* row = ((step * ((mr << 16) - frac)) >> 16) + a*mr + b*xr + c */
#define CRHIP_ROWMODE_UPSAMPLE 1 /* row = (65536 - frac) >> 6          (delta == 0, step == 1024) */

/* Final normalisation accumulator * reciprocal / 32768 (reference clownresampler.h:1033), by proven product range. */
#define CRHIP_NORM_S31 0         /* |acc * reciprocal| < 2^31: signed 24-bit multiply, signed truncating shift */
#define CRHIP_NORM_U32 1         /* |acc| * reciprocal < 2^32: multiply magnitudes unsigned, shift, restore the sign */

/* One launch of the polyphase/LDS kernel: output frames [0, n_out) of a timeline whose frame 0 sits at the
   16.16 position pos0 relative to frame 0 of d_in (the start of the left padding). */
typedef struct crhip_poly_launch
{
	const void *d_in;           /* interleaved int16 */
	uint64_t in_valid_bytes;    /* bytes readable from d_in (loads beyond are suppressed by the buffer descriptor) */
	void *d_out;                /* int32, n_out * channels */
	const int32_t *d_rows;      /* device image of the rows: row_stride/4 planes of plane_rows x 4 int32, swizzled (cr_plan.c) */
	uint64_t pos0;              /* 16.16 */
	uint64_t n_out;
	uint32_t increment;         /* 16.16, < 2^24 */
	uint32_t channels;
	uint32_t slots;             /* taps evaluated per frame */
	uint32_t first_mr;          /* affine row mode: a frame's window starts (min_relative - first_mr) frames after first_slot */
	uint32_t window_extra;      /* largest such shift: the tile's input window holds slots + window_extra frames beyond the last frame's position */
	uint32_t first_slot;        /* frame offset of slot 0 relative to the integer position */
	uint32_t rows;
	uint32_t row_stride;        /* int32 per row, multiple of 4 */
	uint32_t plane_rows;        /* rows rounded up to a multiple of 16 */
	uint32_t swizzle;           /* phys = (row & ~15) | ((row + swizzle * (row >> 4)) & 15) */
	uint32_t row_mode;
	uint32_t norm_mode;
	uint32_t delta;             /* stretched_kernel_radius_delta */
	uint32_t skr;               /* stretched_kernel_radius */
	uint32_t step;              /* kernel_step_size */
	int32_t aff_a, aff_b, aff_c;
	uint32_t threads;           /* workgroup size the instance was compiled for */
	uint32_t vecs;              /* k_poly: 16-byte input vectors per thread per tile (template NV); k_wave: 100 + NVW; k_wave2: 150 + NVW; k_up: 200 */
	uint32_t tile_frames;       /* output frames per tile */
	uint32_t lds_bytes;         /* polyphase rows + two tiles + 16 bytes of mailbox */
	uint32_t blocks;            /* grid size */
	uint32_t specialised;       /* use the (channels, slots) template instance if there is one */
	uint32_t variant;           /* tuning variant of the specialised kernels (CRHIP_VARIANT_DEFAULT = the measured default) */
	uint32_t *d_tickets;        /* CRHIP_TICKET_WORDS zeroed uint32 in device memory: 8 tile-ticket counters and a
	                               finished-workgroup counter, each on its own 128-byte line; the kernel leaves them
	                               zeroed.  Launches that may overlap in time need different blocks */
	unsigned long long *debug_stamps; /* diagnostic instances only: receives {shader cycles, 100 MHz ticks} of workgroup 0 */
	uint32_t dynamic_tiles;     /* k_poly: 1 = tiles beyond the first gridDim.x are drawn as tickets, 0 = plain round-robin */
	uint32_t out_s16;           /* 1: d_out is int16, samples clamped to +-0x7FFF (extension); 0: int32 unclamped (reference) */
	uint32_t wave_tile;         /* k_wave2s: output frames per wave-tile (a multiple of the frames one wave-instruction covers, 64 / ceil(channels / 2)) */
	uint32_t lane_map;          /* k_wave2: 0 = lane l takes frame l of its 64, 1 = lanes 0-31 the even frames, 32-63 the odd ones (LDS bank conflicts of the window reads) */
	/* DUAL MONO (k_poly stereo instances built with DUAL): a MONO stream run as two "channels" - output frame j and output frame
	   j + dual_out_frames, whose fractional positions are equal (dual_out_frames * increment is a multiple of 65536) and which
	   therefore share their row; the second one's window lies dual_in_bytes further on in d_in.  channels is 2, n_out counts
	   PAIRS, d_out is the mono output (int32): pair j writes d_out[j] and - while j < dual_valid_frames - d_out[j + dual_out_frames]. */
	/* 1: the run-time-slot instance computes from PADDED tiles (cr_device.hpp padded_frames; crhip_poly_runtime_padded_frame_bytes says
	   which channel counts have the form): tile_frames is sized for 32-byte frames */
	uint32_t padded;
	uint32_t dual;
	uint32_t dual_out_frames, dual_valid_frames, dual_in_bytes;   /* (all below 2^32: the host launches at most 2^30 pairs) */
	uint32_t debug_form;        /* 0; in a -DCRA_WITH_W2_FORMS build: timing-only forms (results wrong) of k_wave2 (1 ... 3) and of the headline k_poly (1 ... 6) */
} crhip_poly_launch;

/* One launch of the generic kernel: the reference arithmetic restated with 64-bit integers, one thread per
   output frame, weights read from the original table in global memory.  Handles every configuration the
   reference accepts (any channel count up to 16, any ratio), plus the accumulate-into semantics of
   ClownResampler_LowestLevel_Resample (reference clownresampler.h:1020,1033). */
typedef struct crhip_generic_launch
{
	const void *d_in;
	void *d_out;                /* int32 (out64 == 0), int64 (out64 == 1) or clamped int16 (out64 == 2), n_out * channels */
	const int32_t *d_table;     /* the caller's Lanczos table repacked to int32 */
	const int64_t *d_acc_in;    /* optional: channels initial accumulators added to frame 0 (n_out must be 1) */
	uint64_t pos_int, pos_frac; /* of output frame 0 */
	uint64_t increment;
	uint64_t n_out;
	uint64_t skr, radius_frames, delta, step;
	uint32_t table_len;
	uint32_t channels;
	uint32_t out64;
} crhip_generic_launch;

/* One launch of k_int (cr_kint.hpp): whole-number downsampling ratios, increment = ratio << 16.  Every frame of the launch has
   the same fractional position, so the host hands over THE row - staged, in the kernel arguments - instead of a row image:
   w[s] = |weight of slot s| << 15 (plain |weight| for the slots of the instance's safe mask), the signs being a property of
   the instance that the host has checked this row against (crhip_int_instance). */
#define CRHIP_INT_MAX_SLOTS 64
typedef struct crhip_int_launch
{
	const void *d_in;           /* interleaved int16 */
	uint64_t in_valid_bytes;    /* bytes readable from d_in */
	void *d_out;                /* int32 (or clamped int16), n_out * channels */
	uint64_t first_frame;       /* input frame (relative to d_in) that slot 0 of output frame 0 multiplies */
	uint64_t n_out;
	uint32_t channels, ratio, slots;   /* ratio: input frames per PERIOD of `period` output frames (period 1: per output frame) */
	uint32_t period;            /* 1: whole-number ratio; 2, 4: the increment repeats every `period` frames (3:2, 1:2, 1:4 ...) */
	uint32_t out_s16;
	uint32_t blocks;            /* grid size (workgroups of the instance's thread count) */
	int32_t reciprocal[4];      /* per phase: 0x80000000 / sum of the row's weights (clownresampler.h:1025) */
	uint32_t *d_tickets;        /* CRHIP_TICKET_WORDS zeroed counters (as crhip_poly_launch.d_tickets), or NULL: tiles dealt round-robin */
	uint32_t ticket_tiles;      /* wave-tiles per ticket (>= 1): consecutive tiles a wave takes per draw */
	int32_t w[CRHIP_INT_MAX_SLOTS];   /* phase p's slot s at [p * slots + s] */
} crhip_int_launch;

typedef struct crhip_int_shape
{
	uint64_t negmask;           /* bit (p * slots + s) set: slot s of phase p must hold a weight <= 0, clear: >= 0 */
	uint64_t safemask;          /* bit set: that slot may reach 65536; every other weight must stay below it */
	uint64_t zeromask;          /* bit set: that slot's weight must be 0 (the kernel leaves it out) */
	uint32_t period;
	uint32_t starts[4];         /* phase p's first input frame, counted from phase 0's (starts[0] == 0) */
	uint32_t frames_per_lane;   /* K: a wave-tile is 64 K output frames */
	uint32_t threads;           /* workgroup size */
	uint32_t lds_bytes[2];      /* dynamic LDS per workgroup, int32 / int16 output */
} crhip_int_shape;

/* 1 and *shape when there is a k_int instance for (channels, ratio : period, slots), else 0 */
int crhip_int_instance(uint32_t channels, uint32_t ratio, uint32_t period, uint32_t slots, crhip_int_shape *shape);
/* one-time setup (dynamic LDS limit) + workgroups resident per CU for either output form; not legal inside a stream capture */
int crhip_int_prepare(uint32_t channels, uint32_t ratio, uint32_t period, uint32_t slots, int *per_cu, int *per_cu_s16);
int crhip_launch_int(const crhip_int_launch *launch, void *stream);

/* One launch of k_seg (cr_kseg.hpp): long pure-upsampling launches with fixed slot signs (cfg 3's shape).  The 64 lanes of a wave take
   output frames seg_frames apart - seg_frames * increment is a multiple of 65536, so they share their fraction, hence their row,
   which travels in scalar registers.  Lane l of super-block b walks output frames [b * 64 S + l S, + S) (S = seg_frames), K at a
   time (a tile: K frames of each of the 64 segments; tile index = b * tiles_per_seg + t). */
typedef struct crhip_seg_launch
{
	const void *d_in;           /* interleaved int16, stereo */
	uint64_t in_valid_bytes;    /* < 2^32 */
	void *d_out;                /* int32, n_out * 2 */
	const void *d_rows;         /* the FLOAT row image: 16 dwords per row, row-major - |weight| / 65536 as float for the 15 slots, then
	                               2 * reciprocal as int32 (clownresampler.h:1025, :1033) */
	uint64_t pos0;              /* 16.16, of output frame 0 relative to frame 0 of d_in */
	uint64_t n_out;
	uint64_t seg_frames;        /* S: a multiple of 65536 / gcd(increment, 65536) and of tile_frames; 512 S < 2^32 */
	uint64_t seg_in_frames;     /* D = S * increment / 65536; 256 D < 2^32 */
	uint64_t n_tiles;           /* ceil(n_out / (64 S)) * tiles_per_seg, below 2^32 */
	uint32_t increment;         /* 16.16, below 65536 */
	uint32_t first_slot;
	uint32_t slots;             /* 15 */
	uint32_t tile_frames;       /* K: a multiple of the instance's chunk (the last tile of a segment may be shorter) */
	uint32_t tiles_per_seg;     /* ceil(S / K) */
	uint32_t blocks;
	uint32_t *d_tickets;        /* CRHIP_TICKET_WORDS zeroed counters (as crhip_poly_launch.d_tickets) */
	uint32_t debug_form;        /* 0; diagnostic instances: 1-3 timing-only ablations (results wrong), 4 = cycle stamps per phase */
	unsigned long long *debug_stamps;   /* form 4: receives 8 counters per workgroup (cr_kseg.hpp, ABL == 6) */
	uint32_t xcd_run;           /* 0: tiles dealt to the 32 ticket sequences one by one; G (a multiple of 4): in runs of G consecutive tiles per XCD (cr_kseg.hpp) */
} crhip_seg_launch;

/* 1 and the slot signs the instance is built for (as crhip_poly_up_negmask) when there is a k_seg instance for the shape and the ratio;
   *chunk: the frames a lane stages per copy-out (tile_frames must be a multiple) */
int crhip_seg_instance(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode, uint32_t increment, uint32_t *negmask, uint32_t *threads, uint32_t *lds_bytes, uint32_t *chunk);
int crhip_seg_prepare(uint32_t channels, uint32_t slots, uint32_t increment, int *per_cu);   /* one-time setup (dynamic LDS limit); not legal inside a capture */
int crhip_launch_seg(const crhip_seg_launch *launch, void *stream);

/* Many short constant-rate segments of ONE timeline in ONE launch (variable rate: ClownResamplerAMD_ResampleSegmentsDevice): the
   generic kernel's per-frame arithmetic with the segment's parameters looked up in a table - every lane finds its segment by
   bisection over the segments' first output frames. */
typedef struct crhip_segment
{
	uint64_t first_out;         /* index of the segment's first frame in the call's output */
	uint64_t pos_int, pos_frac; /* of that frame, relative to frame 0 of d_in */
	uint64_t increment;
	uint64_t skr, radius_frames, delta, step;   /* the configuration the segment was adjusted to */
} crhip_segment;

typedef struct crhip_segments_launch
{
	const void *d_in;
	void *d_out;                /* int32 or clamped int16, n_out * channels */
	const int32_t *d_table;
	const crhip_segment *d_segments;   /* n_segments entries, first_out ascending, every segment non-empty */
	uint64_t n_out;
	uint32_t n_segments;
	uint32_t table_len;
	uint32_t channels;
	uint32_t out_s16;
} crhip_segments_launch;

int crhip_launch_segments(const crhip_segments_launch *launch, void *stream);

typedef struct crhip_device_info
{
	int compute_units;
	int max_lds_per_block;      /* bytes */
	int wavefront;
	int clock_khz;
	size_t total_memory;
	char name[128];
	char arch[64];
} crhip_device_info;

/* Every function returns 0 on success or a hipError_t value; crhip_error_string translates it. */
const char *crhip_error_string(int code);

int crhip_device_count(int *count);
int crhip_set_device(int ordinal);
int crhip_get_device_info(int ordinal, crhip_device_info *info);

int crhip_malloc(void **device_pointer, size_t bytes);
int crhip_free(void *device_pointer);
int crhip_host_alloc(void **host_pointer, size_t bytes);   /* pinned */
int crhip_host_free(void *host_pointer);
/* 0 and the address the current device reaches [host, host + bytes) under when that range is page-locked host memory (hipHostMalloc,
   hipHostRegister); 1 for pageable memory (nothing is reported, no sticky error is left behind) */
int crhip_host_alias(const void *host, size_t bytes, void **device_alias);
int crhip_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);   /* async on stream */
int crhip_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream);   /* async on stream */
int crhip_memset(void *dst, int value, size_t bytes, void *stream);
int crhip_stream_create(void **stream);
int crhip_stream_destroy(void *stream);
int crhip_stream_sync(void *stream);
int crhip_device_sync(void);
int crhip_event_create(void **event);                   /* timing disabled */
int crhip_event_destroy(void *event);
int crhip_event_record(void *event, void *stream);
int crhip_event_sync(void *event);                      /* the host waits until the event has happened */
int crhip_stream_wait_event(void *stream, void *event);
int crhip_stream_is_capturing(void *stream, int *capturing);   /* *capturing = 1 while the stream records into a graph */
int crhip_stream_busy(void *stream);                    /* 1 = work still in flight, 0 = idle (or the handle is no longer a stream) */
/* peer access + copies between devices (the multi-device entry point's final concatenate) */
int crhip_enable_peer_access(int device, int peer);     /* idempotent; 0 also when the pair has no peer path (copies then stage through the host) */
int crhip_memcpy_peer(void *dst, int dst_device, const void *src, int src_device, size_t bytes, void *stream);
int crhip_get_device(int *ordinal);

#define CRHIP_TICKET_WORDS (33u * 32u)   /* up to 32 ticket counters + the finished counter, 128 bytes apart */

/* variant value meaning: the instance's measured default */
#define CRHIP_VARIANT_DEFAULT 0xFFFFu

/* One-time setup of the instance a launch selects (raises its dynamic-LDS limit); call once per plan, outside any capture. */
int crhip_poly_prepare(const crhip_poly_launch *launch);
int crhip_poly_occupancy(const crhip_poly_launch *launch, int *workgroups_per_cu, int *vgprs, int *static_lds);
int crhip_launch_poly(const crhip_poly_launch *launch, void *stream);
int crhip_launch_generic(const crhip_generic_launch *launch, void *stream);

/* 0, or the bytes a frame takes in the LDS tiles of the PADDED form of the run-time-slot k_poly instance of this channel count (9-11,
   13-15 channels: 32; cr_device.hpp padded_frames) - a launch with `padded` set has its tiles sized with it. */
uint32_t crhip_poly_runtime_padded_frame_bytes(uint32_t channels);
/* 1 when the instance a launch with these parameters selects (launch->dual set) has a dual-mono form */
int crhip_poly_has_dual(const crhip_poly_launch *launch);
/* 1 when a specialised (channels, slots, row mode, norm mode) template instance exists for the polyphase kernel. */
int crhip_poly_has_instance(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode);
/* 1 when the instance a launch with these parameters selects applies the row swizzle. */
int crhip_poly_swizzled(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode, uint32_t variant);
/* measured default of crhip_poly_launch.dynamic_tiles for the instance a launch with these parameters selects */
int crhip_poly_dynamic_default(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode);
/* a k_poly variant to fall back on when the k_wave geometry does not fit a configuration */
uint32_t crhip_poly_fallback_variant(void);
/* k_up (input-stationary upsampling kernel): 1 and *negmask = the per-slot weight signs the instance was built for
   (bit s set: every weight of slot s must be <= 0, clear: >= 0), or 0 when the instance has no k_up form. */
/* ... also the mask of the 64-bit-chain variants of k_poly (variants 28, 29) */
int crhip_poly_up_negmask(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode, uint32_t *negmask);
/* the variant to use when a plan does not qualify for k_up */
/* 1 when the instance's default variant is one of the 64-bit-chain variants (whose sign precondition the host must check) */
int crhip_poly_default_is_mad(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode);
/* 1 when that chain is the any-sign form (downsampling instances): it takes whatever rows the 32-bit kernels take, nothing to check */
int crhip_poly_mad_any_sign(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode);
/* whether the instance has the input-stationary kernel (k_up2, variant 27) / has it as its measured default */
int crhip_poly_has_up(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode);
int crhip_poly_default_is_up(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode);
uint32_t crhip_poly_up_fallback_variant(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode);
/* k_wave2 (expanded window, 64-bit multiply-add taps; variant 30): -1 when the instance has none, 1 when it is built for the slot
   signs in *negmask (the host checks the plan's rows against them, as for k_up), 0 when it takes any rows */
/* k_poly's 64-bit chain (variants 28 / 29) stages every weight outside these slots as |weight| << 15: they must stay below 65536 */
uint32_t crhip_poly_mad_safemask(uint32_t slots);
int crhip_poly_wave2_negmask(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode, uint32_t *negmask);
/* the slots whose weights that k_wave2 instance takes at ANY magnitude up to 65536; every other slot's weights must stay below
   65536 in every row (they are staged as |weight| << 15).  0: no such restriction */
uint32_t crhip_poly_wave2_safemask(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode);
/* the variant to use when a plan does not qualify for k_wave2 */
uint32_t crhip_poly_wave2_fallback_variant(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode);
/* 1 when there is a k_wave2 instance with a run-time slot count for this channel count and row mode (downsampling rows, 1-8
   channels): the host may then give a plan WITHOUT a specialised instance the geometry threads = 64 * waves, vecs = 150 + (1 KiB
   window pieces per wave-tile), tile_frames = 256 and variant CRHIP_VARIANT_RT_WAVE2 */
int crhip_poly_runtime_wave2(uint32_t channels, uint32_t row_mode);
#define CRHIP_VARIANT_RT_WAVE2 31u
/* k_wave2s (cr_kwave2s.hpp): the same arithmetic with one lane per CHANNEL PAIR of a frame - wide frames (9 to 16 channels; fewer
   where the host's rule says so) with long windows.  Geometry: threads = 64 * waves, vecs = 150 + pieces, wave_tile frames per
   wave-tile, tile_frames = 4 * wave_tile, variant CRHIP_VARIANT_RT_WAVE2S.  1 when there is an instance for the channel count. */
int crhip_poly_runtime_wave2s(uint32_t channels);
#define CRHIP_VARIANT_RT_WAVE2S 32u
/* Number of tuning variants of the specialised instances (crhip_poly_launch.variant). */
int crhip_poly_variants(void);
/* Geometry the instance that a launch with these parameters selects is compiled for: workgroup size, 16-byte input
   vectors per thread per tile, and the multiple of output frames a tile should be (threads * frames in flight per lane). */
void crhip_poly_geometry(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode, uint32_t variant,
                         uint32_t *threads, uint32_t *vecs, uint32_t *frames_multiple);

#ifdef __cplusplus
}
#endif

#endif /* CRHIP_H */
