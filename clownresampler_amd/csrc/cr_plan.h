/*
 * cr_plan.h - host-side planning for the GPU hot path (pure C, no HIP): closed forms of the output-timeline
 * walk, and the re-indexing of the caller's Lanczos table into polyphase rows for k_poly (cr_kernels.hip).
 * Internal to libclownresampler_amd.so.
 */
#ifndef CR_PLAN_H
#define CR_PLAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The four scalars of ClownResampler_LowestLevel_Configuration (reference clownresampler.h:632-638) as fixed-width integers. */
typedef struct cr_config
{
	uint64_t skr;           /* stretched_kernel_radius, 16.16 */
	uint64_t radius_frames; /* integer_stretched_kernel_radius */
	uint64_t delta;         /* stretched_kernel_radius_delta, 16.16 */
	uint64_t step;          /* kernel_step_size */
} cr_config;

/* Tap window of one fractional position (reference clownresampler.h:993-1001). */
typedef struct cr_phase
{
	uint32_t first_rel;     /* min_relative: first tap's frame offset from position_integer (padded-buffer frames) */
	uint32_t last_rel;      /* max_relative */
	uint32_t taps;          /* radius_frames + last_rel - first_rel */
	uint64_t table_at;      /* kernel_start */
} cr_phase;

void cr_phase_of(const cr_config *cfg, uint32_t frac, cr_phase *out);

typedef struct cr_poly
{
	int eligible;           /* 1: k_poly may be used */
	const char *reason;     /* why not, when eligible == 0 (static string) */
	int fatal;              /* 1: the reference itself would trap / read outside the table for some phase */

	uint32_t slots;         /* taps evaluated per frame (zero weights included) */
	uint32_t first_slot;    /* frame offset of slot 0 from position_integer (shifted rows: for the phases with the smallest min_relative) */
	uint32_t shifted;       /* 1 (affine row mode): a row's slot 0 is rel_first taps after ITS phase's first tap, i.e. the window of a
	                           frame starts (min_relative - first_mr) frames after first_slot; 0: one window common to all phases */
	uint32_t first_mr;      /* smallest min_relative over all phases */
	uint32_t window_extra;  /* largest shift: frames a tile's input window must hold beyond first_slot + slots */
	uint32_t rel_first;     /* all-zero leading taps trimmed from every row */
	uint32_t rows;
	uint32_t row_stride;    /* int32 per row: [slots weights][reciprocal][zero padding], multiple of 4 */
	uint32_t row_mode;      /* CRHIP_ROWMODE_* */
	int32_t aff_a, aff_b, aff_c;
	uint32_t delta, skr, step;
	uint32_t norm_mode;     /* CRHIP_NORM_*: how the kernel may multiply accumulator and reciprocal */
	int32_t *weights;       /* rows * row_stride, malloc'ed */
} cr_poly;

/* Builds the polyphase rows for (table, cfg).  Walks all 65536 fractional positions, so every row is checked
   against the definition for every position that maps to it.  Returns 0 on success (out->eligible says whether
   the fast kernel's 32-bit preconditions hold), non-zero when out->fatal. */
int cr_poly_build(const int32_t *table, size_t table_len, const cr_config *cfg, cr_poly *out);
void cr_poly_free(cr_poly *poly);
uint32_t cr_poly_row_of(const cr_poly *poly, uint32_t frac);
/* A stream whose increment is PERIODIC (increment * period is a whole number of frames: 3:2, 1:2, ... and period 1 for the
   whole-number ratios) visits `period` fractional positions in turn.  For the `period` consecutive output frames from fractional
   position `frac` on: rows[p] = the row of frame p, starts[p] = the input frame, counted from position_integer of frame 0, that
   slot 0 of frame p multiplies (as row_of / fetch_frame on the device: shifted rows start at their own phase's first tap).
   Returns 0 when a row index falls outside the image. */
int cr_poly_periodic(const cr_poly *poly, uint64_t increment, uint32_t frac, uint32_t period, uint32_t *rows, uint32_t *starts);

/* LDS/device image of the rows: row_stride/4 planes of plane_rows (= rows rounded up to 16) x 4 int32; within each block
   of 16 rows, row r sits at (r & ~15) | ((r + swizzle * (r >> 4)) & 15). */
uint32_t cr_poly_plane_rows(const cr_poly *poly);
uint32_t cr_poly_phys_row(uint32_t row, uint32_t swizzle);
/* Signs of the weights per slot over all rows: bit s of *positive / *negative is set when slot s holds a weight > 0 / < 0
   in some row. */
void cr_poly_slot_signs(const cr_poly *poly, uint32_t *positive, uint32_t *negative);
/* bit s set: some row has a weight of at least this magnitude in slot s */
uint32_t cr_poly_slots_reaching(const cr_poly *poly, int32_t magnitude);
/* Chooses the swizzle (0..15) that minimises ds_read_b128 bank conflicts for lanes that hold consecutive output frames
   `increment` apart; *conflict_cycles_plain / _best receive the modelled extra LDS cycles per wave read. */
uint32_t cr_poly_pick_swizzle(const cr_poly *poly, uint64_t increment, double *conflict_plain, double *conflict_best);
/* ... for a kernel whose lanes take the frames of a block of 64 in another order (crhip_poly_launch.lane_map) */
uint32_t cr_poly_pick_swizzle_mapped(const cr_poly *poly, uint64_t increment, uint32_t lane_map, double *conflict_plain, double *conflict_best);
/* k_wave2's window reads: modelled extra LDS cycles per read of one window slot by a wave, for either lane order */
double cr_window_conflicts(const cr_poly *poly, const cr_config *cfg, uint64_t increment, uint32_t channels, uint32_t lane_map);
/* Device/LDS image of the rows, malloc'ed.  COMPACT (specialised kernels): row_stride/4 planes, a row's int32 [4q,4q+4) in
   plane q, the reciprocal right behind the last weight.  SPLIT (run-time-slot kernels): ceil(slots/4) planes of weights
   (zero-padded) plus one plane holding only the reciprocal.  *device_row_stride receives the int32 per row of the image. */
#define CR_IMAGE_COMPACT 0
#define CR_IMAGE_SPLIT 1
int32_t *cr_poly_device_image(const cr_poly *poly, uint32_t swizzle, int layout, uint32_t *device_row_stride);

/* ---- closed forms of the timeline walk (reference clownresampler.h:1058-1092) ---- */

/* frames emitted while position_integer < total_input_frames */
uint64_t cr_count_output_frames(uint64_t pos_int, uint64_t pos_frac, uint64_t increment, uint64_t total_input_frames);
/* position after `frames` emitted frames */
void cr_advance(uint64_t *pos_int, uint64_t *pos_frac, uint64_t increment, uint64_t frames);
/* number of padded-buffer frames [0, n) that emitting output frames [0, frames) from (pos_int,pos_frac) can read */
uint64_t cr_input_extent(const cr_config *cfg, uint64_t pos_int, uint64_t pos_frac, uint64_t increment, uint64_t frames);

uint64_t cr_hash_bytes(const void *data, size_t bytes, uint64_t seed);

#ifdef __cplusplus
}
#endif

#endif /* CR_PLAN_H */
