/*
 * ref_wrap.c - thin exported wrappers around the REAL reference header, compiled
 * in place from /root/reference (never copied into this repo).
 *
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into
 * oracle/_ref/libclownref_r<R>.so (git-ignored, travels to the GPU box as a
 * prebuilt binary).  It plays the part tests/test-low-level.c:25-28 plays in the
 * reference: one translation unit that defines CLOWNRESAMPLER_IMPLEMENTATION +
 * CLOWNRESAMPLER_STATIC and includes the header, here by the path given on the
 * command line (-DCLOWNREF_HEADER='"/root/reference/clownresampler.h"',
 * -DCLOWNRESAMPLER_KERNEL_RADIUS=<R>).
 *
 * The exported ref_* functions take the same arguments as the oracle_* ones in
 * ../cr_oracle.h (the structs are layout-identical on LP64), so the parity tests
 * can call either library through one ctypes binding.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define CLOWNRESAMPLER_IMPLEMENTATION
#define CLOWNRESAMPLER_STATIC
#include CLOWNREF_HEADER

#define REF_EXPORT __attribute__((visibility("default")))

REF_EXPORT unsigned ref_radius(void) { return CLOWNRESAMPLER_KERNEL_RADIUS; }
REF_EXPORT size_t ref_sizeof_precomputed(void) { return sizeof(ClownResampler_Precomputed); }
REF_EXPORT size_t ref_sizeof_config(void) { return sizeof(ClownResampler_LowestLevel_Configuration); }
REF_EXPORT size_t ref_sizeof_lowlevel(void) { return sizeof(ClownResampler_LowLevel_State); }
REF_EXPORT size_t ref_sizeof_highlevel(void) { return sizeof(ClownResampler_HighLevel_State); }
REF_EXPORT size_t ref_offsetof_lowlevel(int field)
{
	switch (field)
	{
		case 0: return offsetof(ClownResampler_LowLevel_State, lowest_level);
		case 1: return offsetof(ClownResampler_LowLevel_State, channels);
		case 2: return offsetof(ClownResampler_LowLevel_State, position_integer);
		case 3: return offsetof(ClownResampler_LowLevel_State, position_fractional);
		case 4: return offsetof(ClownResampler_LowLevel_State, increment);
	}
	return (size_t)-1;
}
REF_EXPORT size_t ref_offsetof_highlevel(int field)
{
	switch (field)
	{
		case 0: return offsetof(ClownResampler_HighLevel_State, low_level);
		case 1: return offsetof(ClownResampler_HighLevel_State, input_buffer);
		case 2: return offsetof(ClownResampler_HighLevel_State, input_buffer_start);
		case 3: return offsetof(ClownResampler_HighLevel_State, input_buffer_end);
		case 4: return offsetof(ClownResampler_HighLevel_State, maximum_integer_stretched_kernel_radius);
		case 5: return offsetof(ClownResampler_HighLevel_State, leading_padding_frames_needed);
		case 6: return offsetof(ClownResampler_HighLevel_State, trailing_padding_frames_remaining);
	}
	return (size_t)-1;
}

REF_EXPORT size_t ref_table_len(unsigned radius)
{
	(void)radius;
	return CLOWNRESAMPLER_COUNT_OF(((ClownResampler_Precomputed *)0)->lanczos_kernel_table);
}

REF_EXPORT void ref_precompute(int64_t *table, unsigned radius)
{
	(void)radius;
	ClownResampler_Precompute((ClownResampler_Precomputed *)table);
}

REF_EXPORT uint64_t ref_ratio(uint64_t a, uint64_t b)
{
	return ClownResampler_CalculateRatio(a, b);
}

REF_EXPORT uint8_t ref_configure(ClownResampler_LowestLevel_Configuration *cfg, unsigned radius, uint64_t in_rate, uint64_t out_rate, uint64_t lowpass_rate)
{
	(void)radius;
	return ClownResampler_LowestLevel_Configure(cfg, in_rate, out_rate, lowpass_rate);
}

REF_EXPORT void ref_frame(const ClownResampler_LowestLevel_Configuration *cfg, const int64_t *table, size_t table_len, int64_t *accum,
                          uint32_t channels, const int16_t *padded_in, uint64_t pos_int, uint64_t pos_frac)
{
	(void)table_len;
	ClownResampler_LowestLevel_Resample(cfg, (const ClownResampler_Precomputed *)table, (cc_s32f *)accum, channels, padded_in, pos_int, pos_frac);
}

REF_EXPORT uint8_t ref_low_init(ClownResampler_LowLevel_State *st, unsigned radius, uint32_t channels, uint64_t in_rate, uint64_t out_rate, uint64_t lowpass_rate)
{
	(void)radius;
	return ClownResampler_LowLevel_Init(st, channels, in_rate, out_rate, lowpass_rate);
}

REF_EXPORT uint8_t ref_low_adjust(ClownResampler_LowLevel_State *st, unsigned radius, uint64_t in_rate, uint64_t out_rate, uint64_t lowpass_rate)
{
	(void)radius;
	return ClownResampler_LowLevel_Adjust(st, in_rate, out_rate, lowpass_rate);
}

REF_EXPORT uint8_t ref_low_resample(ClownResampler_LowLevel_State *st, const int64_t *table, size_t table_len, const int16_t *padded_in,
                                    size_t *frames_left, ClownResampler_OutputCallback emit, const void *user)
{
	(void)table_len;
	return ClownResampler_LowLevel_Resample(st, (const ClownResampler_Precomputed *)table, padded_in, frames_left, emit, user);
}

REF_EXPORT uint8_t ref_high_init(ClownResampler_HighLevel_State *st, unsigned radius, uint32_t channels, uint64_t in_rate, uint64_t out_rate, uint64_t lowpass_rate)
{
	(void)radius;
	return ClownResampler_HighLevel_Init(st, channels, in_rate, out_rate, lowpass_rate);
}

REF_EXPORT uint8_t ref_high_resample(ClownResampler_HighLevel_State *st, const int64_t *table, size_t table_len, ClownResampler_InputCallback pull,
                                     ClownResampler_OutputCallback emit, const void *user)
{
	(void)table_len;
	return ClownResampler_HighLevel_Resample(st, (const ClownResampler_Precomputed *)table, pull, emit, user);
}

REF_EXPORT uint8_t ref_high_adjust(ClownResampler_HighLevel_State *st, unsigned radius, uint64_t in_rate, uint64_t out_rate, uint64_t lowpass_rate)
{
	(void)radius;
	return ClownResampler_HighLevel_Adjust(st, in_rate, out_rate, lowpass_rate);
}

REF_EXPORT uint8_t ref_high_end(ClownResampler_HighLevel_State *st, const int64_t *table, size_t table_len, ClownResampler_OutputCallback emit, const void *user)
{
	(void)table_len;
	return ClownResampler_HighLevel_ResampleEnd(st, (const ClownResampler_Precomputed *)table, emit, user);
}

/* ---- the same harness conveniences as cr_oracle.c, driving the reference ---- */

typedef struct store_ctx
{
	int32_t *out;
	size_t written;
	size_t capacity;
} store_ctx;

static cc_bool store_emit(void *user, const cc_s32f *frame, cc_u8f samples)
{
	store_ctx *ctx = (store_ctx *)user;
	int32_t *dst = ctx->out + ctx->written * samples;
	cc_u8f c;

	for (c = 0; c < samples; ++c)
		dst[c] = (int32_t)(uint32_t)(unsigned long)frame[c];

	return ++ctx->written < ctx->capacity;
}

REF_EXPORT size_t ref_low_resample_i32(ClownResampler_LowLevel_State *st, const int64_t *table, size_t table_len, const int16_t *padded_in,
                                       size_t *frames_left, int32_t *out, size_t out_capacity_frames, int norm_mode,
                                       uint64_t legacy_gain, uint8_t *ran_out_of_input)
{
	store_ctx ctx;
	cc_bool exhausted;

	(void)table_len;
	(void)legacy_gain;

	if (norm_mode != 0)
		return (size_t)-1; /* the shipped header has only the current normalisation */

	ctx.out = out;
	ctx.written = 0;
	ctx.capacity = out_capacity_frames;

	if (out_capacity_frames == 0 && st->position_integer < *frames_left)
	{
		if (ran_out_of_input != NULL)
			*ran_out_of_input = 0;
		return 0;
	}

	exhausted = ClownResampler_LowLevel_Resample(st, (const ClownResampler_Precomputed *)table, padded_in, frames_left, store_emit, &ctx);

	if (ran_out_of_input != NULL)
		*ran_out_of_input = exhausted;

	return ctx.written;
}

typedef struct pcm_source
{
	const int16_t *pcm;
	size_t frames_left;
	size_t chunk;
	unsigned channels;
	store_ctx sink;
} pcm_source;

static size_t pcm_pull(void *user, cc_s16l *buffer, size_t max_frames)
{
	pcm_source *src = (pcm_source *)user;
	size_t n = max_frames < src->frames_left ? max_frames : src->frames_left;

	if (src->chunk != 0 && n > src->chunk)
		n = src->chunk;

	memcpy(buffer, src->pcm, n * src->channels * sizeof(*buffer));
	src->pcm += n * src->channels;
	src->frames_left -= n;
	return n;
}

static cc_bool pcm_emit(void *user, const cc_s32f *frame, cc_u8f samples)
{
	pcm_source *src = (pcm_source *)user;
	return store_emit(&src->sink, frame, samples);
}

REF_EXPORT size_t ref_high_run_i32(ClownResampler_HighLevel_State *st, const int64_t *table, size_t table_len, const int16_t *pcm,
                                   size_t pcm_frames, size_t pull_chunk, int32_t *out, size_t out_capacity_frames)
{
	pcm_source src;

	(void)table_len;

	src.pcm = pcm;
	src.frames_left = pcm_frames;
	src.chunk = pull_chunk;
	src.channels = st->low_level.channels;
	src.sink.out = out;
	src.sink.written = 0;
	src.sink.capacity = out_capacity_frames;

	if (ClownResampler_HighLevel_Resample(st, (const ClownResampler_Precomputed *)table, pcm_pull, pcm_emit, &src))
		ClownResampler_HighLevel_ResampleEnd(st, (const ClownResampler_Precomputed *)table, pcm_emit, &src);

	return src.sink.written;
}
