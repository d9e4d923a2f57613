/*
 * cr_api.c - one instance of the clownresampler API per kernel radius.
 *
 * Compiled once per supported CLOWNRESAMPLER_KERNEL_RADIUS (see Makefile); include/clownresampler.h redirects
 * the function names to ..._R<radius> for radii other than the default 3, so all instances live in one library.
 *
 * Host C.  What stays on the host is what the reference runs once per stream or per call: the table
 * (clownresampler.h:892-961), the ratio/configuration scalars (:913-984, :1044-1056), the high-level API's
 * staging-buffer choreography (:1101-1250) and the bookkeeping of the output-timeline walk (:1058-1092).
 * The per-frame arithmetic (:986-1035) is NOT implemented here: every frame comes from the HIP kernels behind
 * cr_context.c, and when no device is usable these functions end in the error handler.
 */
#include "../../include/clownresampler_amd.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "cr_context.h"

#define RADIUS CLOWNRESAMPLER_KERNEL_RADIUS
#define TABLE_LEN ((size_t)RADIUS * 2u * CLOWNRESAMPLER_KERNEL_RESOLUTION)
#define ONE 65536ul /* 16.16 */

#if CLOWNRESAMPLER_KERNEL_RADIUS != 3
 #define ClownResamplerAMD_BuildRows CLOWNRESAMPLER_AMD_SYM(ClownResamplerAMD_BuildRows)
 #define ClownResamplerAMD_PeriodicShape CLOWNRESAMPLER_AMD_SYM(ClownResamplerAMD_PeriodicShape)
#endif

/* ======================================================================================================= */
/* Table                                                                                                   */
/* ======================================================================================================= */

void ClownResampler_Precompute(ClownResampler_Precomputed *precomputed)
{
	/* Lanczos-windowed sinc sampled at TABLE_LEN points over [-R, R), in 16.16, truncated toward zero.
	   The order of the floating-point operations follows the reference (clownresampler.h:894-907, :960) because
	   the integer table must come out identical: x*pi, (x*pi)/R, sin*sin/(a*b). */
	const double radius = (double)RADIUS;
	const double pi = 3.14159265358979323846264338327950288;
	size_t i;

	for (i = 0; i < TABLE_LEN; ++i)
	{
		const double x = ((double)i / (double)TABLE_LEN * 2.0 - 1.0) * radius;
		double value = 1.0;

		if (x != 0.0)
		{
			const double a = x * pi;
			const double b = a / radius;
			value = (sin(a) * sin(b)) / (a * b);
		}

		precomputed->lanczos_kernel_table[i] = (cc_s32l)(value * (double)ONE);
	}
}

/* ======================================================================================================= */
/* Ratio and configuration                                                                                 */
/* ======================================================================================================= */

/* floor(numerator * 65536 / denominator) in arithmetic that never needs more than the width of cc_u32f:
   three base-65536 digits divided from the top, the remainder carried down (clownresampler.h:913-953).
   0xFFFFFFFF marks a zero operand or a quotient of 2^32 or more; a zero quotient becomes 1. */
static cc_u32f ratio_16_16(cc_u32f numerator, cc_u32f denominator)
{
	cc_u32f digit[3];
	cc_u32f quotient;

	if (numerator == 0 || denominator == 0)
		return 0xFFFFFFFF;

	digit[2] = numerator / ONE;
	digit[1] = numerator % ONE;
	digit[0] = 0;

	digit[1] |= digit[2] % denominator * ONE;
	digit[2] /= denominator;
	digit[0] |= digit[1] % denominator * ONE;
	digit[1] /= denominator;
	digit[0] /= denominator;

	if (digit[2] != 0 || digit[1] >= ONE)
		return 0xFFFFFFFF;

	quotient = digit[1] * ONE + digit[0];
	return quotient != 0 ? quotient : 1;
}

cc_bool ClownResampler_LowestLevel_Configure(ClownResampler_LowestLevel_Configuration *configuration, cc_u32f input_sample_rate, cc_u32f output_sample_rate, cc_u32f low_pass_filter_sample_rate)
{
	/* the kernel is only ever stretched, to the lowest of the three rates (clownresampler.h:965-970) */
	cc_u32f pass_rate = input_sample_rate;
	cc_u32f stretch, squeeze;

	if (output_sample_rate < pass_rate)
		pass_rate = output_sample_rate;
	if (low_pass_filter_sample_rate < pass_rate)
		pass_rate = low_pass_filter_sample_rate;

	stretch = ratio_16_16(input_sample_rate, pass_rate);
	squeeze = ratio_16_16(pass_rate, input_sample_rate);

	if (stretch >= 0x1000 * ONE) /* :974, leaves *configuration untouched */
		return cc_false;

	configuration->stretched_kernel_radius = RADIUS * stretch;                                                                    /* :977 */
	configuration->integer_stretched_kernel_radius = (configuration->stretched_kernel_radius + (ONE - 1)) / ONE;                   /* :978 */
	configuration->stretched_kernel_radius_delta = configuration->integer_stretched_kernel_radius * ONE - configuration->stretched_kernel_radius; /* :979 */
	configuration->kernel_step_size = CLOWNRESAMPLER_KERNEL_RESOLUTION * squeeze / ONE;                                           /* :981 */

	return cc_true;
}

cc_bool ClownResampler_LowLevel_Adjust(ClownResampler_LowLevel_State *resampler, cc_u32f input_sample_rate, cc_u32f output_sample_rate, cc_u32f low_pass_filter_sample_rate)
{
	/* increment first, validation second: a failed Adjust leaves the new increment behind (clownresampler.h:1054-1055) */
	resampler->increment = ratio_16_16(input_sample_rate, output_sample_rate);
	return ClownResampler_LowestLevel_Configure(&resampler->lowest_level, input_sample_rate, output_sample_rate, low_pass_filter_sample_rate);
}

cc_bool ClownResampler_LowLevel_Init(ClownResampler_LowLevel_State *resampler, cc_u8f channels, cc_u32f input_sample_rate, cc_u32f output_sample_rate, cc_u32f low_pass_filter_sample_rate)
{
	resampler->channels = channels; /* not range-checked, as in the reference (clownresampler.h:1046) */
	resampler->position_integer = 0;
	resampler->position_fractional = 0;
	return ClownResampler_LowLevel_Adjust(resampler, input_sample_rate, output_sample_rate, low_pass_filter_sample_rate);
}

/* ======================================================================================================= */
/* Plans                                                                                                   */
/* ======================================================================================================= */

static int fill_table_i32(const void *user, int32_t *dst, size_t count)
{
	const ClownResampler_Precomputed *precomputed = (const ClownResampler_Precomputed *)user;
	size_t i;

	for (i = 0; i < count; ++i)
	{
		const long long v = precomputed->lanczos_kernel_table[i];   /* cc_s32l: long (C89 types) or int_least32_t */

		if (v < -2147483647LL - 1 || v > 2147483647LL)
			return -1;

		dst[i] = (int32_t)v;
	}

	return 0;
}

static void config_of(const ClownResampler_LowestLevel_Configuration *configuration, cr_config *cfg)
{
	cfg->skr = configuration->stretched_kernel_radius;
	cfg->radius_frames = configuration->integer_stretched_kernel_radius;
	cfg->delta = configuration->stretched_kernel_radius_delta;
	cfg->step = configuration->kernel_step_size;
}

/* Plans are keyed by the CONTENTS of the caller's table (48 KB to hash): a caller may legitimately keep several
   tables, or const-initialise one from a dump (clownresampler.h:677-681). */
static uint64_t table_hash_of(const ClownResampler_Precomputed *precomputed)
{
	return cr_hash_bytes(precomputed->lanczos_kernel_table, sizeof(precomputed->lanczos_kernel_table), RADIUS);
}

/* The caller holds the plan until cr_plan_release. */
static ClownResamplerAMD_Plan *plan_for_hashed(uint64_t table_hash, const ClownResampler_LowestLevel_Configuration *configuration, const ClownResampler_Precomputed *precomputed, cc_u8f channels, cc_u32f increment, int pin)
{
	cr_config cfg;

	config_of(configuration, &cfg);
	return cr_plan_get(table_hash, TABLE_LEN, fill_table_i32, precomputed, RADIUS, &cfg, channels, increment, pin);
}

static ClownResamplerAMD_Plan *plan_for(const ClownResampler_LowestLevel_Configuration *configuration, const ClownResampler_Precomputed *precomputed, cc_u8f channels, cc_u32f increment, int pin)
{
	return plan_for_hashed(table_hash_of(precomputed), configuration, precomputed, channels, increment, pin);
}

ClownResamplerAMD_Plan *ClownResamplerAMD_PlanCreate(const ClownResampler_LowLevel_State *state, const ClownResampler_Precomputed *precomputed)
{
	/* pinned: valid until ClownResamplerAMD_Shutdown, as the header promises */
	ClownResamplerAMD_Plan *plan = plan_for(&state->lowest_level, precomputed, state->channels, state->increment, 1);

	cr_plan_release(plan);
	return plan;
}

/* Host-only view of the polyphase rows a plan would use (no device needed): for tests and tools.
   *rows_out is malloc'ed (info->rows * info->row_stride int32); free() it.  Returns 0, or non-zero with
   *reason set when the configuration is unusable even for the reference. */
int ClownResamplerAMD_BuildRows(const ClownResampler_LowestLevel_Configuration *configuration, const ClownResampler_Precomputed *precomputed,
                                ClownResamplerAMD_PlanInfo *info, int32_t **rows_out, int *eligible, const char **reason, uint32_t *row_of_fraction)
{
	cr_config cfg;
	cr_poly poly;
	int32_t *table = (int32_t *)malloc(TABLE_LEN * sizeof(int32_t));
	int r;
	uint32_t frac;

	config_of(configuration, &cfg);
	memset(info, 0, sizeof(*info));
	*rows_out = NULL;
	*eligible = 0;
	*reason = "";

	if (table == NULL || fill_table_i32(precomputed, table, TABLE_LEN) != 0)
	{
		free(table);
		*reason = "table entry does not fit 32 bits";
		return -1;
	}

	r = cr_poly_build(table, TABLE_LEN, &cfg, &poly);
	free(table);
	*reason = poly.reason;
	*eligible = poly.eligible;
	info->slots = poly.slots;
	info->first_slot = poly.first_slot;
	info->rows = poly.rows;
	info->row_stride = poly.row_stride;
	info->row_mode = poly.row_mode;
	info->norm_mode = poly.norm_mode;
	*rows_out = poly.weights; /* ownership passes to the caller */

	/* optional: the row every one of the 65536 fractional positions maps to (host mirror of the device formula) */
	if (row_of_fraction != NULL && poly.weights != NULL)
		for (frac = 0; frac < 65536u; ++frac)
			row_of_fraction[frac] = cr_poly_row_of(&poly, frac);

	return r;
}

/* Host-only: what a k_int instance for this configuration and (periodic) increment has to be compiled for - the period, the input
   frames per period, every phase's first input frame (counted from phase 0's), and per phase-slot whether the weight is <= 0 and
   whether it reaches 65536 - at fractional position 0.  tools/int_shapes.py prints the instance table's constants with it. */
int ClownResamplerAMD_PeriodicShape(const ClownResampler_LowestLevel_Configuration *configuration, const ClownResampler_Precomputed *precomputed,
                                    uint64_t increment, uint32_t *period_out, uint32_t *ratio_out, uint32_t *slots_out, uint32_t starts_out[4],
                                    uint64_t *negmask_out, uint64_t *safemask_out, uint64_t *zeromask_out)
{
	cr_config cfg;
	cr_poly poly;
	int32_t *table = (int32_t *)malloc(TABLE_LEN * sizeof(int32_t));
	uint32_t period, rows[4], starts[4], p, s;
	int ok = -1;

	config_of(configuration, &cfg);
	if (table == NULL || fill_table_i32(precomputed, table, TABLE_LEN) != 0)
	{
		free(table);
		return -1;
	}
	if (cr_poly_build(table, TABLE_LEN, &cfg, &poly) != 0 || poly.weights == NULL)
	{
		free(table);
		cr_poly_free(&poly);
		return -1;
	}
	free(table);
	for (period = 1; period <= 4u; period *= 2u)
		if (((increment * period) & 0xFFFFu) == 0)
			break;
	if (period <= 4u && poly.slots * period <= 64u && cr_poly_periodic(&poly, increment, 0, period, rows, starts))
	{
		*period_out = period;
		*ratio_out = (uint32_t)((increment * period) >> 16);
		*slots_out = poly.slots;
		*negmask_out = *safemask_out = *zeromask_out = 0;
		for (p = 0; p < period; ++p)
		{
			const int32_t *w = poly.weights + (size_t)rows[p] * poly.row_stride;

			starts_out[p] = starts[p] - starts[0];
			for (s = 0; s < poly.slots; ++s)
			{
				if (w[s] < 0)
					*negmask_out |= 1ull << (p * poly.slots + s);
				if (w[s] >= 65536 || w[s] <= -65536)
					*safemask_out |= 1ull << (p * poly.slots + s);
				if (w[s] == 0)
					*zeromask_out |= 1ull << (p * poly.slots + s);
			}
		}
		ok = 0;
	}
	cr_poly_free(&poly);
	return ok;
}

/* ======================================================================================================= */
/* The output-timeline walk (clownresampler.h:1058-1092), bulk form                                        */
/* ======================================================================================================= */

/* Leaves state and *total_input_frames as the reference does after `emitted` frames, the last of which made the
   consumer say stop (clownresampler.h:1084-1088). */
static void settle_stopped(ClownResampler_LowLevel_State *resampler, size_t *total_input_frames, uint64_t pos_int, uint64_t pos_frac, uint64_t emitted)
{
	size_t consumed;

	cr_advance(&pos_int, &pos_frac, resampler->increment, emitted);
	consumed = pos_int < *total_input_frames ? (size_t)pos_int : *total_input_frames;
	*total_input_frames -= consumed;
	resampler->position_integer = (size_t)pos_int - consumed;
	resampler->position_fractional = (cc_u32f)pos_frac;
}

/* ... and after the input ran out (clownresampler.h:1065-1067). */
static void settle_exhausted(ClownResampler_LowLevel_State *resampler, size_t *total_input_frames, uint64_t pos_int, uint64_t pos_frac, uint64_t emitted)
{
	cr_advance(&pos_int, &pos_frac, resampler->increment, emitted);
	resampler->position_integer = (size_t)pos_int - *total_input_frames;
	resampler->position_fractional = (cc_u32f)pos_frac;
	*total_input_frames = 0;
}

static size_t resample_bulk(ClownResampler_LowLevel_State *resampler, const ClownResampler_Precomputed *precomputed, const cc_s16l *input_buffer, size_t *total_input_frames, void *output, size_t output_capacity_frames, cc_bool *ran_out_of_input, int out_s16)
{
	const uint64_t pos_int = resampler->position_integer, pos_frac = resampler->position_fractional;
	const uint64_t available = cr_count_output_frames(pos_int, pos_frac, resampler->increment, *total_input_frames);
	const uint64_t emit = available < output_capacity_frames ? available : output_capacity_frames;
	/* the consumer "returns 0" on the frame that fills it, even if that is also the last frame there is */
	const int stopped = available >= output_capacity_frames && available != 0;

	if (ran_out_of_input != NULL)
		*ran_out_of_input = stopped ? cc_false : cc_true;

	if (emit != 0)
	{
		const ClownResamplerAMD_Plan *plan = plan_for(&resampler->lowest_level, precomputed, resampler->channels, resampler->increment, 0);
		int failed;

		/* a failure (recorded: ClownResamplerAMD_LastErrorCode) = "the consumer took nothing and said stop": state and *total_input_frames untouched */
		if (plan == NULL)
		{
			if (ran_out_of_input != NULL)
				*ran_out_of_input = cc_false;
			return 0;
		}

		failed = cr_run_host(plan, input_buffer, (uint64_t)*total_input_frames + 2 * resampler->lowest_level.integer_stretched_kernel_radius,
		                     pos_int, pos_frac, emit, output, out_s16);
		cr_plan_release(plan);
		if (failed != 0)
		{
			if (ran_out_of_input != NULL)
				*ran_out_of_input = cc_false;
			return 0;
		}
	}
	else if (stopped)
	{
		return 0; /* no room for even one frame: nothing happens, as if the call had not been made */
	}

	if (stopped)
		settle_stopped(resampler, total_input_frames, pos_int, pos_frac, emit);
	else
		settle_exhausted(resampler, total_input_frames, pos_int, pos_frac, emit);

	return (size_t)emit;
}

size_t ClownResampler_LowLevel_ResampleBulk(ClownResampler_LowLevel_State *resampler, const ClownResampler_Precomputed *precomputed, const cc_s16l *input_buffer, size_t *total_input_frames, int32_t *output, size_t output_capacity_frames, cc_bool *ran_out_of_input)
{
	return resample_bulk(resampler, precomputed, input_buffer, total_input_frames, output, output_capacity_frames, ran_out_of_input, 0);
}

size_t ClownResampler_LowLevel_ResampleBulkS16(ClownResampler_LowLevel_State *resampler, const ClownResampler_Precomputed *precomputed, const cc_s16l *input_buffer, size_t *total_input_frames, int16_t *output, size_t output_capacity_frames, cc_bool *ran_out_of_input)
{
	return resample_bulk(resampler, precomputed, input_buffer, total_input_frames, output, output_capacity_frames, ran_out_of_input, 1);
}

/* ======================================================================================================= */
/* Variable rate on the device: a list of constant-rate segments, one launch each, no synchronisation      */
/* ======================================================================================================= */

/* Equivalent reference sequence, per segment s covering input frames [a_s, a_s + n_s) of one contiguous timeline:
     ClownResampler_LowLevel_Adjust(state, rates_s);                                   clownresampler.h:1052-1056
     total = n_s;  ClownResampler_LowLevel_Resample(state, precomputed,                clownresampler.h:1058-1092
                       timeline + (a_s - integer_stretched_kernel_radius_s) * channels, &total, append, 0);
   i.e. each segment's "padding" is the real neighbouring frames (legal per clownresampler.h:725-733), the position
   (overshoot + fraction) is carried across the re-configuration exactly as the state struct carries it. */
/* One launch for all segments from this many on, if a segment averages less WORK than that - output frames x channels x taps.  A
   launch per segment costs ~6 us each on either side of the bus (43 us where the segment's ratio needs a new plan), the generic
   kernel ~2.8 ps per tap and sample: 600 segments of 44,100 stereo frames took 26 ms one by one and 0.8 ms in one launch,
   6 of 4.4 M frames 0.09 against 0.5 ms (profiles/r03_segments_one_launch.log). */
#define SEGMENT_TABLE_MIN_SEGMENTS 8u
#define SEGMENT_TABLE_MAX_WORK 2000000u

size_t ClownResamplerAMD_ResampleSegmentsDevice(ClownResampler_LowLevel_State *resampler, const ClownResampler_Precomputed *precomputed,
                                                const void *device_timeline, size_t halo_frames, const ClownResamplerAMD_Segment *segments, size_t segment_count,
                                                void *device_output, size_t output_capacity_frames, int output_is_s16, size_t *segment_output_frames, void *hip_stream)
{
	const unsigned long errors_before = cr_error_serial();
	ClownResampler_LowLevel_State state = *resampler;
	uint64_t total_out = 0, in_frame = 0, out_frame = 0, table_hash;
	size_t s, table_count = 0;
	crhip_segment *table = NULL;
	const ClownResamplerAMD_Plan *table_plan = NULL;
	int use_table = 0;

	/* pass 1, host only: every segment must be acceptable before anything is enqueued */
	for (s = 0; s < segment_count; ++s)
	{
		uint64_t n, pos_int, pos_frac;

		if (!ClownResampler_LowLevel_Adjust(&state, segments[s].input_sample_rate, segments[s].output_sample_rate, segments[s].low_pass_filter_sample_rate))
		{
			cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "segment %lu: ClownResampler_LowLevel_Adjust rejects the rates %lu -> %lu (low-pass %lu)", (unsigned long)s,
			        (unsigned long)segments[s].input_sample_rate, (unsigned long)segments[s].output_sample_rate, (unsigned long)segments[s].low_pass_filter_sample_rate);
			return 0;
		}
		if (state.lowest_level.integer_stretched_kernel_radius > halo_frames)
		{
			cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "segment %lu needs %lu frames of halo, the timeline has %lu", (unsigned long)s,
			        (unsigned long)state.lowest_level.integer_stretched_kernel_radius, (unsigned long)halo_frames);
			return 0;
		}

		pos_int = state.position_integer;
		pos_frac = state.position_fractional;
		n = cr_count_output_frames(pos_int, pos_frac, state.increment, segments[s].input_frames);
		cr_advance(&pos_int, &pos_frac, state.increment, n);
		state.position_integer = (size_t)(pos_int - segments[s].input_frames);     /* :1065 */
		state.position_fractional = (cc_u32f)pos_frac;
		total_out += n;
	}

	if (total_out > output_capacity_frames)
	{
		cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "the segments produce %lu frames, the output has room for %lu", (unsigned long)total_out, (unsigned long)output_capacity_frames);
		return 0;
	}

	/* pass 2.  Long segments: one launch each, back to back on the caller's stream, on whatever kernel the segment's plan has.  MANY
	   SHORT ones: a launch costs ~5 us however little it does (6,000 segments of a tenth of a second: 31 ms), so they all go into ONE
	   launch of the generic kernel with a segment table (k_generic_segments: ~0.1 ns per frame slower, no per-segment cost). */
	table_hash = table_hash_of(precomputed);
	state = *resampler;
	{
		size_t non_empty = 0;
		uint64_t taps = 1;
		ClownResampler_LowLevel_State probe = *resampler;

		for (s = 0; s < segment_count; ++s)
		{
			uint64_t pi, pf, n;

			ClownResampler_LowLevel_Adjust(&probe, segments[s].input_sample_rate, segments[s].output_sample_rate, segments[s].low_pass_filter_sample_rate);
			pi = probe.position_integer;
			pf = probe.position_fractional;
			n = cr_count_output_frames(pi, pf, probe.increment, segments[s].input_frames);
			non_empty += n != 0;
			if (n != 0 && 2u * probe.lowest_level.integer_stretched_kernel_radius > taps)
				taps = 2u * probe.lowest_level.integer_stretched_kernel_radius;
			cr_advance(&pi, &pf, probe.increment, n);
			probe.position_integer = (size_t)(pi - segments[s].input_frames);
			probe.position_fractional = (cc_u32f)pf;
		}
		use_table = cr_segments_mode() == 2 || (cr_segments_mode() == 0 && non_empty >= SEGMENT_TABLE_MIN_SEGMENTS
		                                            && total_out / non_empty * probe.channels * taps < SEGMENT_TABLE_MAX_WORK);
		/* Never on a stream that is recording into a hipGraph: the table travels through ONE pinned buffer and ONE device buffer
		   per device that the next call overwrites, behind an event this call records - a graph would replay the copy of whatever the
		   buffer holds by then, and the recorded event never "happens" for the host.  Launches per segment carry everything in their
		   kernel arguments and own never-recycled ticket blocks under capture (cr_context.c, capture pool): they replay correctly. */
		if (use_table)
		{
			int capturing = 0;

			if (hip_stream != NULL && (crhip_stream_is_capturing(hip_stream, &capturing) != 0 || capturing))
				use_table = 0;
		}
		if (use_table && non_empty != 0)
		{
			table = (crhip_segment *)malloc(non_empty * sizeof(crhip_segment));
			if (table == NULL)
			{
				cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "out of host memory");
				return 0;
			}
		}
	}
	for (s = 0; s < segment_count; ++s)
	{
		uint64_t n;
		size_t total = segments[s].input_frames;

		ClownResampler_LowLevel_Adjust(&state, segments[s].input_sample_rate, segments[s].output_sample_rate, segments[s].low_pass_filter_sample_rate);
		n = cr_count_output_frames(state.position_integer, state.position_fractional, state.increment, total);

		if (n != 0 && use_table)
		{
			/* positions relative to the timeline's frame 0 - radius: the window of a segment starts radius_frames before its first frame */
			crhip_segment *e = &table[table_count++];
			const size_t radius_frames = state.lowest_level.integer_stretched_kernel_radius;

			e->first_out = out_frame;
			e->pos_int = (uint64_t)halo_frames + in_frame - radius_frames + state.position_integer;
			e->pos_frac = state.position_fractional;
			e->increment = state.increment;
			e->skr = state.lowest_level.stretched_kernel_radius;
			e->radius_frames = radius_frames;
			e->delta = state.lowest_level.stretched_kernel_radius_delta;
			e->step = state.lowest_level.kernel_step_size;
			if (table_plan == NULL)
			{
				table_plan = plan_for_hashed(table_hash, &state.lowest_level, precomputed, state.channels, state.increment, 0);
				if (table_plan == NULL)
				{
					free(table);
					return 0;
				}
			}
		}
		else if (n != 0)
		{
			const size_t radius_frames = state.lowest_level.integer_stretched_kernel_radius;
			const cc_s16l *window = (const cc_s16l *)device_timeline + ((ptrdiff_t)in_frame - (ptrdiff_t)radius_frames) * (ptrdiff_t)state.channels;
			unsigned char *out = (unsigned char *)device_output + out_frame * state.channels * (output_is_s16 ? sizeof(int16_t) : sizeof(int32_t));
			const ClownResamplerAMD_Plan *plan = plan_for_hashed(table_hash, &state.lowest_level, precomputed, state.channels, state.increment, 0);
			int failed;

			if (plan == NULL)
				return 0;
			/* released as soon as the launch is enqueued: rows are only ever freed with hipFree, which waits for it */
			failed = cr_plan_launch(plan, window, ((uint64_t)total + 2 * radius_frames) * state.channels * sizeof(cc_s16l), out,
			                        state.position_integer, state.position_fractional, n, hip_stream, output_is_s16);
			cr_plan_release(plan);
			if (failed != 0)
				return 0;
		}

		settle_exhausted(&state, &total, state.position_integer, state.position_fractional, n);
		if (segment_output_frames != NULL)
			segment_output_frames[s] = (size_t)n;
		in_frame += segments[s].input_frames;
		out_frame += n;
	}
	if (table_plan != NULL)
	{
		/* d_in of the table launch: `halo_frames` frames before the timeline's frame 0 (the caller's buffer starts there) */
		const cc_s16l *base = (const cc_s16l *)device_timeline - (ptrdiff_t)halo_frames * (ptrdiff_t)state.channels;
		const int failed = cr_segments_run(table_plan, base, device_output, table, table_count, out_frame, output_is_s16, hip_stream);

		cr_plan_release(table_plan);
		free(table);
		if (failed != 0)
			return 0;
	}
	else
		free(table);

	if (cr_error_serial() != errors_before)
		return 0;

	*resampler = state;
	return (size_t)out_frame;
}

/* ======================================================================================================= */
/* Several GPUs, one call (cr_multi.c)                                                                      */
/* ======================================================================================================= */

size_t ClownResamplerAMD_ResampleShardedDevice(ClownResampler_LowLevel_State *resampler, const ClownResampler_Precomputed *precomputed, size_t total_input_frames,
                                               const ClownResamplerAMD_DeviceShard *shards, unsigned shard_count, int output_is_s16,
                                               int gather_mode, unsigned root_shard, void *root_output)
{
	return cr_resample_sharded(resampler, table_hash_of(precomputed), TABLE_LEN, fill_table_i32, precomputed, RADIUS, total_input_frames,
	                           shards, shard_count, output_is_s16, gather_mode, root_shard, root_output);
}

/* ======================================================================================================= */
/* The callback form (the reference's own signature)                                                       */
/* ======================================================================================================= */

#ifndef CLOWNRESAMPLER_NO_LOW_LEVEL_API

/* Hands frames [0, n) of `batch_out` to the consumer, the state moved on BEFORE each frame is handed out
   (clownresampler.h:1076-1081: a callback that looks at the state sees what it would see with the reference).  Returns the
   number of frames handed out; *stopped = 1 when the callback returned 0 on the last of them (which may be frame n - 1). */
static uint64_t replay_frames(ClownResampler_LowLevel_State *resampler, const int32_t *batch_out, uint64_t n, ClownResampler_OutputCallback output_callback, const void *user_data, int *stopped)
{
	const cc_u8f channels = resampler->channels;
	const uint64_t increment = resampler->increment;
	/* the running position in locals (the callback cannot be assumed not to touch memory, so fields of *resampler would be
	   reloaded after every call); the struct is brought up to date before every call */
	uint64_t position = ((uint64_t)resampler->position_integer << 16) + resampler->position_fractional;
	uint64_t i;

	for (i = 0; i < n; ++i)
	{
		cc_s32f frame[CLOWNRESAMPLER_MAXIMUM_CHANNELS];
		const int32_t *from = batch_out + i * channels;
		cc_u8f c;

		if (channels == 2)
		{
			frame[0] = from[0];
			frame[1] = from[1];
		}
		else
			for (c = 0; c < channels; ++c)
				frame[c] = from[c];

		position += increment;
		resampler->position_integer = (size_t)(position >> 16);
		resampler->position_fractional = (cc_u32f)(position & (ONE - 1u));

		if (!output_callback((void *)user_data, frame, channels))
		{
			*stopped = 1;
			return i + 1;
		}
	}
	*stopped = 0;
	return n;
}

/* The consumer's callbacks run on the calling thread, one frame at a time, at 1-2 ns each: for a long call that - not the GPU -
   is where the time goes (10 minutes of stereo: ~45 ms of callbacks against ~5 ms of upload + kernel + download).  So once the
   batches have reached their full size, a helper thread computes batch k + 1 while the calling thread replays batch k (two
   slots).  A consumer that stops early costs at most the batch in flight, as before.  The helper only computes: callbacks AND
   error reports happen on the calling thread (a device failure over there is handed back and raised here).
   CLOWNRESAMPLER_AMD_NO_REPLAY_THREAD=1 in the environment keeps the whole call on the calling thread. */
typedef struct replay_ahead
{
	const ClownResamplerAMD_Plan *plan;
	const cc_s16l *input_buffer;
	uint64_t padded_frames, start_int, start_frac, increment;
	uint64_t first, available, batch;   /* output frames [first, available), `batch` at a time */
	int32_t *slot[2];
	uint64_t frames[2];
	int state[2];                        /* 0: free, 1: filled, 2: the device failed (error_code / error_message: re-raised by the calling thread) */
	int cancel;
	int error_code;
	char error_message[512];
	pthread_mutex_t lock;
	pthread_cond_t changed;
} replay_ahead;

static void *replay_ahead_worker(void *argument)
{
	replay_ahead *a = (replay_ahead *)argument;
	uint64_t at = a->first;
	unsigned k = 0;

	/* failures on this thread are recorded, not reported: the client's error handler runs on the client's thread only */
	cr_error_defer(1);

	while (at < a->available)
	{
		const uint64_t n = a->available - at < a->batch ? a->available - at : a->batch;
		uint64_t pi = a->start_int, pf = a->start_frac;
		int failed;

		pthread_mutex_lock(&a->lock);
		while (a->state[k] != 0 && !a->cancel)
			pthread_cond_wait(&a->changed, &a->lock);
		if (a->cancel)
		{
			pthread_mutex_unlock(&a->lock);
			break;
		}
		pthread_mutex_unlock(&a->lock);

		cr_advance(&pi, &pf, a->increment, at);
		failed = cr_run_host(a->plan, a->input_buffer, a->padded_frames, pi, pf, n, a->slot[k], 0) != 0;

		pthread_mutex_lock(&a->lock);
		if (failed)
			a->error_code = cr_error_take(a->error_message, sizeof(a->error_message));
		a->frames[k] = n;
		a->state[k] = failed ? 2 : 1;
		pthread_cond_broadcast(&a->changed);
		pthread_mutex_unlock(&a->lock);
		if (failed)
			break;
		at += n;
		k ^= 1u;
	}
	return NULL;
}

cc_bool ClownResampler_LowLevel_Resample(ClownResampler_LowLevel_State *resampler, const ClownResampler_Precomputed *precomputed, const cc_s16l *input_buffer, size_t *total_input_frames, ClownResampler_OutputCallback output_callback, const void *user_data)
{
	const uint64_t start_int = resampler->position_integer, start_frac = resampler->position_fractional;
	const uint64_t available = cr_count_output_frames(start_int, start_frac, resampler->increment, *total_input_frames);
	const cc_u8f channels = resampler->channels;
	const uint64_t padded_frames = (uint64_t)*total_input_frames + 2 * resampler->lowest_level.integer_stretched_kernel_radius;
	const ClownResamplerAMD_Plan *plan;
	int32_t *batch_out;
	uint64_t done = 0;
	int failed = 0, consumer_stopped = 0;
	/* A consumer may stop after a handful of frames (a sound-card callback asks for a few hundred,
	   examples/low-level.c:84) or take the whole stream: frames are computed ahead speculatively in batches
	   that start small and grow, so neither case wastes much. */
	uint64_t batch = 1024;
	const uint64_t batch_limit = (1u << 21) / (channels != 0 ? channels : 1u);   /* 8 MiB of int32 per batch */

	if (available == 0)
	{
		settle_exhausted(resampler, total_input_frames, start_int, start_frac, 0);
		return cc_true;
	}

	/* A failure - no plan, no memory, a launch or a copy that the device refuses - is reported as the reference's OTHER outcome: "the
	   callback said stop" (cc_false), the state and *total_input_frames as they are after the frames the consumer HAS been given (none:
	   untouched).  The reference cannot fail here (clownresampler.h:746-748); of its two outcomes this is the one after which a client
	   still owns its unprocessed input.  ClownResamplerAMD_LastErrorCode() tells the two apart. */
	plan = plan_for(&resampler->lowest_level, precomputed, channels, resampler->increment, 0);
	if (plan == NULL)
		return cc_false;

	batch_out = (int32_t *)malloc((size_t)(available < batch_limit ? available : batch_limit) * channels * sizeof(int32_t));
	if (batch_out == NULL)
	{
		cr_plan_release(plan);
		cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "out of host memory");
		return cc_false;
	}

	/* 1. the growing batches, one after the other on this thread */
	while (done < available && !(batch == batch_limit && available - done > 2 * batch_limit))
	{
		const uint64_t n = available - done < batch ? available - done : batch;
		uint64_t handed;

		{
			uint64_t pi = start_int, pf = start_frac;
			cr_advance(&pi, &pf, resampler->increment, done);
			if (cr_run_host(plan, input_buffer, padded_frames, pi, pf, n, batch_out, 0) != 0)
			{
				failed = 1;
				break;
			}
		}

		handed = replay_frames(resampler, batch_out, n, output_callback, user_data, &consumer_stopped);
		if (consumer_stopped)
		{
			free(batch_out);
			cr_plan_release(plan);
			settle_stopped(resampler, total_input_frames, start_int, start_frac, done + handed);
			return cc_false;
		}

		done += n;
		if (batch < batch_limit)
			batch *= 4;
		if (batch > batch_limit)
			batch = batch_limit;
	}

	/* 2. what is left of a long call, at full batch size: computed one batch ahead of the callbacks by a helper thread */
	if (!failed && done < available)
	{
		replay_ahead ahead;
		pthread_t worker;
		int32_t *second = (int32_t *)malloc((size_t)batch_limit * channels * sizeof(int32_t));
		unsigned k = 0;
		int stopped = 0, started = 0;

		memset(&ahead, 0, sizeof(ahead));
		ahead.plan = plan;
		ahead.input_buffer = input_buffer;
		ahead.padded_frames = padded_frames;
		ahead.start_int = start_int;
		ahead.start_frac = start_frac;
		ahead.increment = resampler->increment;
		ahead.first = done;
		ahead.available = available;
		ahead.batch = batch_limit;
		ahead.slot[0] = batch_out;
		ahead.slot[1] = second;
		pthread_mutex_init(&ahead.lock, NULL);
		pthread_cond_init(&ahead.changed, NULL);
		started = second != NULL && !cr_env_no_replay_thread() && pthread_create(&worker, NULL, replay_ahead_worker, &ahead) == 0;

		while (started && done < available)
		{
			uint64_t n, handed;
			int state;

			pthread_mutex_lock(&ahead.lock);
			while (ahead.state[k] == 0)
				pthread_cond_wait(&ahead.changed, &ahead.lock);
			state = ahead.state[k];
			n = ahead.frames[k];
			pthread_mutex_unlock(&ahead.lock);
			if (state == 2)
			{
				/* the helper's failure, reported HERE: the calling thread's error state, the client's handler on the client's thread */
				cr_fail(ahead.error_code != 0 ? ahead.error_code : CLOWNRESAMPLER_AMD_ERROR_HIP, "%s", ahead.error_message[0] != '\0' ? ahead.error_message : "the device failed in the compute-ahead thread");
				failed = 1;
				break;
			}

			handed = replay_frames(resampler, ahead.slot[k], n, output_callback, user_data, &consumer_stopped);
			if (consumer_stopped)
			{
				done += handed;
				stopped = 1;
				break;
			}
			done += n;

			pthread_mutex_lock(&ahead.lock);
			ahead.state[k] = 0;
			pthread_cond_broadcast(&ahead.changed);
			pthread_mutex_unlock(&ahead.lock);
			k ^= 1u;
		}

		if (started)
		{
			pthread_mutex_lock(&ahead.lock);
			ahead.cancel = 1;
			pthread_cond_broadcast(&ahead.changed);
			pthread_mutex_unlock(&ahead.lock);
			pthread_join(worker, NULL);
		}
		pthread_cond_destroy(&ahead.changed);
		pthread_mutex_destroy(&ahead.lock);
		free(second);

		if (stopped)
		{
			free(batch_out);
			cr_plan_release(plan);
			settle_stopped(resampler, total_input_frames, start_int, start_frac, done);
			return cc_false;
		}

		/* (no helper thread or no memory for its second slot: the rest one batch after the other, as above) */
		while (!started && !failed && done < available)
		{
			const uint64_t n = available - done < batch_limit ? available - done : batch_limit;
			uint64_t pi = start_int, pf = start_frac, handed;

			cr_advance(&pi, &pf, resampler->increment, done);
			if (cr_run_host(plan, input_buffer, padded_frames, pi, pf, n, batch_out, 0) != 0)
			{
				failed = 1;
				break;
			}
			handed = replay_frames(resampler, batch_out, n, output_callback, user_data, &consumer_stopped);
			if (consumer_stopped)
			{
				free(batch_out);
				cr_plan_release(plan);
				settle_stopped(resampler, total_input_frames, start_int, start_frac, done + handed);
				return cc_false;
			}
			done += n;
		}
	}

	free(batch_out);
	cr_plan_release(plan);

	if (failed || done < available)
	{
		/* a device failure was recorded (and the handler, if any, returned): stop where we are - as if the consumer had said stop on the
		   last frame it was given; if it was given none, nothing has happened */
		if (done != 0)
			settle_stopped(resampler, total_input_frames, start_int, start_frac, done);
		return cc_false;
	}

	settle_exhausted(resampler, total_input_frames, start_int, start_frac, available);
	return cc_true;
}
#endif

/* ======================================================================================================= */
/* One frame (clownresampler.h:986-1035)                                                                   */
/* ======================================================================================================= */

void ClownResampler_LowestLevel_Resample(const ClownResampler_LowestLevel_Configuration *configuration, const ClownResampler_Precomputed *precomputed, cc_s32f *output_frame, cc_u8f channels, const cc_s16l *input_buffer, size_t position_integer, cc_u32f position_fractional)
{
	/* increment is irrelevant for a single frame; 65536 keeps the plan key well-formed */
	const ClownResamplerAMD_Plan *plan = plan_for(configuration, precomputed, channels, ONE, 0);
	int64_t acc_in[CLOWNRESAMPLER_MAXIMUM_CHANNELS], acc_out[CLOWNRESAMPLER_MAXIMUM_CHANNELS];
	cc_u8f c;

	if (plan == NULL)
		return;

	for (c = 0; c < channels; ++c)
		acc_in[c] = output_frame[c];

	/* the frame reads padded-buffer frames [position_integer, position_integer + 2 * radius) at most (:995-996, :1003-1004) */
	{
		const int failed = cr_run_single_frame(plan, input_buffer + position_integer * channels, 2 * (uint64_t)configuration->integer_stretched_kernel_radius,
		                                       position_fractional, acc_in, acc_out);
		cr_plan_release(plan);
		if (failed != 0)
			return;
	}

	for (c = 0; c < channels; ++c)
		output_frame[c] = (cc_s32f)acc_out[c];
}

/* ======================================================================================================= */
/* High-level API (clownresampler.h:1101-1250): host-side staging around the low-level call                */
/* ======================================================================================================= */

#ifndef CLOWNRESAMPLER_NO_HIGH_LEVEL_API

#define STAGING_SAMPLES CLOWNRESAMPLER_COUNT_OF(((ClownResampler_HighLevel_State *)0)->input_buffer)

/* The state's own 0x1000-sample buffer (clownresampler.h:654) is not used for audio here: the staging lives in a side
   window that can grow (cr_stream, cr_context.c), registered under the state's ADDRESS.  The buffer's first bytes carry the
   key of that window, so the state stays a plain struct the caller owns.  Like the reference's state - whose
   input_buffer_start / _end point into itself (clownresampler.h:655-656) - it cannot be moved by a byte copy: the key is
   only honoured at the address it was issued for.  ClownResampler_HighLevel_Init never reads the caller's (possibly
   uninitialised) struct to find a window; it asks the registry for the one of that address. */
#define STREAM_MAGIC 0x314C48444D415243ull /* "CRAMDHL1" */

static void stream_key_store(ClownResampler_HighLevel_State *resampler, uint64_t id)
{
	const uint64_t magic = STREAM_MAGIC;

	memcpy((unsigned char *)resampler->input_buffer, &magic, sizeof(magic));
	memcpy((unsigned char *)resampler->input_buffer + sizeof(magic), &id, sizeof(id));
}

static cr_stream *stream_of(ClownResampler_HighLevel_State *resampler)
{
	uint64_t magic, id;

	memcpy(&magic, (unsigned char *)resampler->input_buffer, sizeof(magic));
	memcpy(&id, (unsigned char *)resampler->input_buffer + sizeof(magic), sizeof(id));
	return magic == STREAM_MAGIC ? cr_stream_lookup(id, resampler) : NULL;
}

/* frames the reference asks its input callback for per refill (clownresampler.h:1154) */
static size_t reference_pull_frames(const ClownResampler_HighLevel_State *resampler)
{
	return (STAGING_SAMPLES - 2 * resampler->maximum_integer_stretched_kernel_radius * resampler->low_level.channels) / resampler->low_level.channels;
}

cc_bool ClownResampler_HighLevel_Init(ClownResampler_HighLevel_State *resampler, cc_u8f channels, cc_u32f input_sample_rate, cc_u32f output_sample_rate, cc_u32f low_pass_filter_sample_rate)
{
	cr_stream *stream;
	size_t halo_samples;

	if (channels > CLOWNRESAMPLER_MAXIMUM_CHANNELS) /* :1103 */
		return cc_false;

	if (!ClownResampler_LowLevel_Init(&resampler->low_level, channels, input_sample_rate, output_sample_rate, low_pass_filter_sample_rate))
		return cc_false;

	/* A window wider than the reference's staging buffer: ClownResampler_HighLevel_Adjust refuses it (:1202), the reference's Init
	   does not look - and its first refill then asks the input callback for (0x1000 - 2 * radius * channels) / channels frames, a
	   negative count as a size_t, into a buffer with no room at all: undefined behaviour (found by tests/soak_gpu.py: 6 channels
	   43 -> 3 Hz with a 1 Hz filter, 13 channels 192 -> 8 kHz at radius 8).  Here Init applies Adjust's rule and refuses. */
	if (channels != 0 && resampler->low_level.lowest_level.integer_stretched_kernel_radius * 2 >= STAGING_SAMPLES / channels)
		return cc_false;

	/* the radius at Init is the largest this state will ever accept (:1109, :1195) */
	resampler->maximum_integer_stretched_kernel_radius = resampler->low_level.lowest_level.integer_stretched_kernel_radius;
	resampler->leading_padding_frames_needed = resampler->maximum_integer_stretched_kernel_radius;
	resampler->trailing_padding_frames_remaining = resampler->maximum_integer_stretched_kernel_radius;

	/* silence where the frames before the stream would be; empty window right behind it (:1112-1115) */
	halo_samples = resampler->maximum_integer_stretched_kernel_radius * channels;
	/* re-initialising a state - or initialising one at an address a discarded state used - reuses that address's window
	   (the reference has no Deinit; ClownResamplerAMD_HighLevel_Release frees a window explicitly) */
	memset(resampler->input_buffer, 0, sizeof(resampler->input_buffer));
	stream = cr_stream_claim(resampler);

	if (stream == NULL || channels == 0 || cr_stream_reserve(stream, 2 * halo_samples + reference_pull_frames(resampler) * channels) != 0)
	{
		cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, channels == 0 ? "channel count 0" : "out of host memory");
		return cc_false;
	}

	memset(stream->window, 0, halo_samples * sizeof(stream->window[0]));
	stream->start = stream->end = halo_samples;
	stream->target_frames = reference_pull_frames(resampler);
	stream_key_store(resampler, stream->id);
	resampler->input_buffer_start = resampler->input_buffer_end = NULL; /* the window may move; see cr_stream */

	return cc_true;
}

/* reference_schedule: refill with ONE pull of the reference's size, as the reference does (:1154) - for ClownResampler_HighLevel_ResampleEnd,
   whose input callback is the library's own and counts trailing_padding_frames_remaining down in the caller's state: with a read-ahead
   window a flush stopped by the consumer would otherwise have taken more of the padding than the reference's (radius 145, 11 channels:
   82 frames per pull - the reference's state says 63 left where one big refill said 0; found by tests/soak_gpu.py).  A flush is at
   most `radius` frames: nothing to read ahead for. */
static cc_bool high_level_resample(ClownResampler_HighLevel_State *resampler, const ClownResampler_Precomputed *precomputed, ClownResampler_InputCallback input_callback, ClownResampler_OutputCallback output_callback, const void *user_data, int reference_schedule)
{
	const size_t channels = resampler->low_level.channels;
	const size_t halo_samples = resampler->maximum_integer_stretched_kernel_radius * channels;
	const size_t one_pull = reference_pull_frames(resampler);
	cr_stream *stream = stream_of(resampler);
	cc_bool consumer_stopped = cc_false;

	if (stream == NULL)
	{
		cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "ClownResampler_HighLevel_Resample on a state that ClownResampler_HighLevel_Init of this library did not set up");
		return cc_true;
	}

	/* Window layout, as the reference's staging buffer: [ left halo | frames ... | look-ahead ], the newest `halo` frames
	   not yet consumed.  First, collect the look-ahead: the first real frames go right-aligned into [halo, 2*halo)
	   (:1127-1136). */
	while (resampler->leading_padding_frames_needed != 0)
	{
		cc_s16l *where = stream->window + 2 * halo_samples - resampler->leading_padding_frames_needed * channels;
		const size_t got = input_callback((void *)user_data, where, resampler->leading_padding_frames_needed);

		if (got == 0)
			return cc_true;

		resampler->leading_padding_frames_needed -= got;
	}

	while (!consumer_stopped)
	{
		if (stream->start == stream->end)
		{
			/* window used up: the last 2*halo samples (left halo + look-ahead of the next window) move to the front and
			   new frames are pulled in behind them (:1143-1158).  The reference pulls ONCE per refill, into what is
			   left of its 0x1000 samples; here the same-sized pulls are repeated until the window's target is reached
			   or the source runs dry, so that one GPU call covers them all. */
			size_t limit = stream->target_frames;
			size_t have = 0;

			if (limit > cr_stream_max_frames())
				limit = cr_stream_max_frames();
			if (limit < one_pull || reference_schedule)
				limit = one_pull;

			memmove(stream->window, stream->window + stream->end - halo_samples, 2 * halo_samples * sizeof(stream->window[0]));

			if (cr_stream_reserve(stream, 2 * halo_samples + limit * channels) != 0)
			{
				/* (the window keeps its two halos, empty in between: the next call refills) */
				stream->start = stream->end = halo_samples;
				cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "out of host memory");
				return cc_false;
			}

			stream->start = halo_samples;
			stream->pull_count = 0;

			while (have + one_pull <= limit || have == 0)
			{
				const size_t got = input_callback((void *)user_data, stream->window + 2 * halo_samples + have * channels, one_pull);

				have += got;

				if (got == 0)
					break;

				if (cr_stream_note_pull(stream, stream->start + have * channels) != 0)
				{
					/* (the frames pulled so far stay in the window: the next call resamples them) */
					stream->end = stream->start + have * channels;
					cr_fail(CLOWNRESAMPLER_AMD_ERROR_ARGUMENT, "out of host memory");
					return cc_false;
				}
			}

			stream->end = stream->start + have * channels;

			if (have == 0)
				return cc_true;
		}

		{
			/* the CURRENT radius flanks the window: it may be smaller than the one at Init after an Adjust (:1165) */
			const size_t current_halo = resampler->low_level.lowest_level.integer_stretched_kernel_radius * channels;
			size_t frames = (stream->end - stream->start) / channels;
			const unsigned long errors_before = cr_error_serial();

			consumer_stopped = !ClownResampler_LowLevel_Resample(&resampler->low_level, precomputed, stream->window + stream->start - current_halo, &frames, output_callback, user_data);

			if (consumer_stopped && stream->pull_count > 1 && cr_error_serial() == errors_before)
			{
				/* The window holds several of the reference's windows (one per pull), and what the reference's state says after a
				   consumer's stop depends on which of them it was in: it consumes whole frames up to the next frame's position
				   but never beyond the END OF THAT WINDOW (:1085-1088, delta = min(position, frames of the window)), the rest of
				   the way staying in position_integer.  ClownResampler_LowLevel_Resample above has applied that rule to the whole
				   big window; redo it with the pull's end, so that the caller-visible state (and the frame the next
				   ClownResampler_HighLevel_Adjust takes effect at - the same either way) is the reference's for any window. */
				const size_t old_start = stream->start;
				const size_t consumed = (stream->end - old_start) / channels - frames;
				const size_t next_position = consumed + resampler->low_level.position_integer;   /* of the next frame, from old_start */
				/* the frame the consumer stopped at: one increment back (16.16; :1076-1078) */
				const uint64_t next_fixed = ((uint64_t)next_position << 16) + (uint64_t)resampler->low_level.position_fractional;
				const size_t stopped_at = (size_t)((next_fixed - (uint64_t)resampler->low_level.increment) >> 16);
				size_t k = 0, window_end, take;

				while (k + 1 < stream->pull_count && stream->pull_ends[k] <= old_start + stopped_at * channels)
					++k;

				window_end = (stream->pull_ends[k] - old_start) / channels;   /* frames from old_start to the end of that pull */
				take = next_position < window_end ? next_position : window_end;
				resampler->low_level.position_integer = next_position - take;
				frames = (stream->end - old_start) / channels - take;
			}

			stream->start = stream->end - frames * channels; /* :1171 */

			/* a device failure was recorded: what the consumer has not been given stays in the window (the next call goes on there),
			   and the caller gets the "output callback said stop" outcome - looping here would never end */
			if (cr_error_serial() != errors_before)
				return cc_false;

			/* read further ahead only for consumers that take everything they are given */
			if (consumer_stopped)
				stream->target_frames = one_pull;
			else if (stream->target_frames < cr_stream_max_frames())
				stream->target_frames *= 4;
		}
	}

	return cc_false;
}

cc_bool ClownResampler_HighLevel_Resample(ClownResampler_HighLevel_State *resampler, const ClownResampler_Precomputed *precomputed, ClownResampler_InputCallback input_callback, ClownResampler_OutputCallback output_callback, const void *user_data)
{
	return high_level_resample(resampler, precomputed, input_callback, output_callback, user_data, 0);
}

#ifndef CLOWNRESAMPLER_NO_HIGH_LEVEL_ADJUST
cc_bool ClownResampler_HighLevel_Adjust(ClownResampler_HighLevel_State *resampler, cc_u32f input_sample_rate, cc_u32f output_sample_rate, cc_u32f low_pass_filter_sample_rate)
{
	const ClownResampler_LowLevel_State before = resampler->low_level;
	const cc_bool accepted = ClownResampler_LowLevel_Adjust(&resampler->low_level, input_sample_rate, output_sample_rate, low_pass_filter_sample_rate)
	    /* a wider kernel than at Init would outgrow the halos already in the staging buffer (:1195) */
	    && resampler->low_level.lowest_level.integer_stretched_kernel_radius <= resampler->maximum_integer_stretched_kernel_radius
	    /* and both halos must leave room in the staging buffer (:1202) */
	    && resampler->low_level.lowest_level.integer_stretched_kernel_radius * 2 < STAGING_SAMPLES / resampler->low_level.channels;

	if (!accepted)
		resampler->low_level = before;

	return accepted;
}
#endif

#ifndef CLOWNRESAMPLER_NO_HIGH_LEVEL_RESAMPLE_END
typedef struct drain_context
{
	ClownResampler_HighLevel_State *resampler;
	ClownResampler_OutputCallback output_callback;
	void *user_data;
} drain_context;

/* feeds what is left of the trailing silence (:1223-1233) */
static size_t drain_input(void *user_data, cc_s16l *buffer, size_t total_frames)
{
	drain_context *context = (drain_context *)user_data;
	size_t frames = context->resampler->trailing_padding_frames_remaining;

	if (frames > total_frames)
		frames = total_frames;

	memset(buffer, 0, frames * context->resampler->low_level.channels * sizeof(*buffer));
	context->resampler->trailing_padding_frames_remaining -= frames;
	return frames;
}

static cc_bool drain_output(void *user_data, const cc_s32f *frame, cc_u8f total_samples)
{
	drain_context *context = (drain_context *)user_data;
	return context->output_callback(context->user_data, frame, total_samples);
}

cc_bool ClownResampler_HighLevel_ResampleEnd(ClownResampler_HighLevel_State *resampler, const ClownResampler_Precomputed *precomputed, ClownResampler_OutputCallback output_callback, const void *user_data)
{
	/* flushes the look-ahead frames by pushing `radius` frames of silence through the same path (:1242-1250) */
	drain_context context;

	context.resampler = resampler;
	context.output_callback = output_callback;
	context.user_data = (void *)user_data;

	return high_level_resample(resampler, precomputed, drain_input, drain_output, &context, 1);
}
#endif

#endif /* CLOWNRESAMPLER_NO_HIGH_LEVEL_API */
