/*
 * cr_oracle.c - CPU restatement of clownresampler's windowed-sinc path.
 *
 * TEST INFRASTRUCTURE ONLY - see cr_oracle.h for who may use this and how its
 * parity with the real reference is pinned.
 *
 * Every function cites the reference lines (/root/reference/clownresampler.h)
 * whose BEHAVIOUR it restates.  The arithmetic is kept scalar, one output
 * frame at a time, one integer divide per frame and one indirect call per
 * frame, so that timing it is representative of the reference C path
 * (BASELINE.md section 4).  It is written from the behavioural description in
 * SURVEY.md, with 64-bit integers spelled out where the reference relies on
 * LP64 `long`.
 */
#include "cr_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* Table (clownresampler.h:892-908, :955-961)                                */
/* ------------------------------------------------------------------------- */

size_t oracle_table_len(unsigned radius)
{
	/* clownresampler.h:629 - radius * 2 * resolution entries */
	return (size_t)radius * 2u * ORACLE_TABLE_RES;
}

/* Lanczos window times sinc, evaluated in the reference's operation order
 * (clownresampler.h:894-907): x*pi, (x*pi)/R, sin*sin/(a*b); L(0) = 1. */
static double lanczos_at(double x, double radius)
{
	/* the literal of clownresampler.h:896 rounds to this double */
	const double a = x * 3.14159265358979323846264338327950288;
	const double b = a / radius;

	if (x == 0.0)
		return 1.0;

	return (sin(a) * sin(b)) / (a * b);
}

void oracle_precompute(int64_t *table, unsigned radius)
{
	const size_t n = oracle_table_len(radius);
	size_t i;

	/* clownresampler.h:959-960: argument ((i/n)*2 - 1)*R, scaled by 65536, C cast (truncates toward zero) */
	for (i = 0; i < n; ++i)
	{
		const double x = ((double)i / (double)n * 2.0 - 1.0) * (double)radius;
		table[i] = (int64_t)(lanczos_at(x, (double)radius) * (double)ORACLE_FRAC_ONE);
	}
}

/* ------------------------------------------------------------------------- */
/* Ratio and configuration (clownresampler.h:913-953, :963-984)              */
/* ------------------------------------------------------------------------- */

/* floor(a * 65536 / b) by base-65536 long division over three limbs, with the
 * reference's sentinels: 0xFFFFFFFF for a zero operand (clownresampler.h:919)
 * or overflow (:939), and 1 in place of 0 (:949). */
uint64_t oracle_ratio(uint64_t a, uint64_t b)
{
	uint64_t limb_hi, limb_mid, limb_lo, q;

	if (a == 0 || b == 0)
		return 0xFFFFFFFFu;

	/* a * 65536 as limbs [hi:mid:lo] = [a / 65536 : a % 65536 : 0]  (clownresampler.h:924-926) */
	limb_hi = a / ORACLE_FRAC_ONE;
	limb_mid = a % ORACLE_FRAC_ONE;
	limb_lo = 0;

	/* schoolbook division, remainder carried down one limb at a time (:929-936);
	   the carries are OR-ed in, as the reference does */
	limb_mid |= limb_hi % b * ORACLE_FRAC_ONE;
	limb_hi /= b;
	limb_lo |= limb_mid % b * ORACLE_FRAC_ONE;
	limb_mid /= b;
	limb_lo /= b;

	if (limb_hi != 0 || limb_mid >= ORACLE_FRAC_ONE)
		return 0xFFFFFFFFu;

	q = limb_mid * ORACLE_FRAC_ONE + limb_lo;

	return q == 0 ? 1 : q;
}

uint8_t oracle_configure(oracle_config *cfg, unsigned radius, uint64_t in_rate, uint64_t out_rate, uint64_t lowpass_rate)
{
	/* clownresampler.h:968-970: the pass band is the lowest of the three rates */
	uint64_t band = lowpass_rate < out_rate ? lowpass_rate : out_rate;
	uint64_t stretch, squeeze;

	if (in_rate < band)
		band = in_rate;

	stretch = oracle_ratio(in_rate, band);
	squeeze = oracle_ratio(band, in_rate);

	/* clownresampler.h:974 - refuse a stretch of 4096 or more; cfg untouched */
	if (stretch >= (uint64_t)0x1000 * ORACLE_FRAC_ONE)
		return 0;

	/* clownresampler.h:977-981 */
	cfg->stretched_radius = (uint64_t)radius * stretch;
	cfg->radius_frames = (cfg->stretched_radius + (ORACLE_FRAC_ONE - 1)) / ORACLE_FRAC_ONE;
	cfg->radius_delta = cfg->radius_frames * ORACLE_FRAC_ONE - cfg->stretched_radius;
	cfg->table_step = (uint64_t)ORACLE_TABLE_RES * squeeze / ORACLE_FRAC_ONE;

	return 1;
}

/* ------------------------------------------------------------------------- */
/* One output frame (clownresampler.h:986-1035)                              */
/* ------------------------------------------------------------------------- */

static inline __attribute__((always_inline)) void frame_core(const oracle_config *cfg, const int64_t *table, size_t table_len, int64_t *accum, uint32_t channels,
                       const int16_t *padded_in, uint64_t pos_int, uint64_t pos_frac, int norm_mode, uint64_t legacy_gain)
{
	/* tap window relative to pos_int, in padded-buffer frames (clownresampler.h:993-996) */
	const uint64_t first_rel = (pos_frac + cfg->radius_delta + (ORACLE_FRAC_ONE - 1)) / ORACLE_FRAC_ONE;
	const uint64_t last_rel = (pos_frac + cfg->stretched_radius) / ORACLE_FRAC_ONE;
	const uint64_t first_frame = pos_int + first_rel;
	const uint64_t end_frame = pos_int + cfg->radius_frames + last_rel;

	/* table index of the first tap: step * (first_rel - frac) in 16.16, floored (clownresampler.h:1001) */
	uint64_t table_at = cfg->table_step * (first_rel * ORACLE_FRAC_ONE - pos_frac) / ORACLE_FRAC_ONE;

	int64_t weight_sum = 0;
	uint64_t frame;
	uint32_t c;

	(void)table_len;

	for (frame = first_frame; frame < end_frame; ++frame, table_at += cfg->table_step)
	{
		const int64_t weight = table[table_at];                    /* :1015 */
		const int16_t *src = padded_in + frame * channels;

		weight_sum += weight;                                       /* :1016 */

		/* every tap is truncated toward zero BEFORE it is accumulated (:1020) */
		for (c = 0; c < channels; ++c)
			accum[c] += (int64_t)src[c] * weight / ORACLE_FRAC_ONE;
	}

	if (norm_mode == ORACLE_NORM_LEGACY_GAIN)
	{
		/* not in the shipped header: see ORACLE_NORM_LEGACY_GAIN in cr_oracle.h */
		for (c = 0; c < channels; ++c)
			accum[c] = accum[c] * (int64_t)legacy_gain / ORACLE_FRAC_ONE;
	}
	else
	{
		/* 17.15 reciprocal of the weight sum, one exact signed divide per frame (:1025),
		   then a truncating 17.15 multiply of the WHOLE accumulator (:1033) */
		const int64_t recip = (int64_t)0x80000000u / weight_sum;

		for (c = 0; c < channels; ++c)
			accum[c] = accum[c] * recip / 32768;
	}
}

void oracle_frame(const oracle_config *cfg, const int64_t *table, size_t table_len, int64_t *accum, uint32_t channels,
                  const int16_t *padded_in, uint64_t pos_int, uint64_t pos_frac)
{
	frame_core(cfg, table, table_len, accum, channels, padded_in, pos_int, pos_frac, ORACLE_NORM_CURRENT, 0);
}

/* ------------------------------------------------------------------------- */
/* Low-level API (clownresampler.h:1044-1092)                                */
/* ------------------------------------------------------------------------- */

uint8_t oracle_low_adjust(oracle_lowlevel *st, unsigned radius, uint64_t in_rate, uint64_t out_rate, uint64_t lowpass_rate)
{
	/* the increment is stored before the configuration is validated (clownresampler.h:1054-1055) */
	st->increment = oracle_ratio(in_rate, out_rate);
	return oracle_configure(&st->cfg, radius, in_rate, out_rate, lowpass_rate);
}

uint8_t oracle_low_init(oracle_lowlevel *st, unsigned radius, uint32_t channels, uint64_t in_rate, uint64_t out_rate, uint64_t lowpass_rate)
{
	/* clownresampler.h:1046-1049; channels is not range-checked here */
	st->channels = channels;
	st->pos_int = 0;
	st->pos_frac = 0;
	return oracle_low_adjust(st, radius, in_rate, out_rate, lowpass_rate);
}

static inline __attribute__((always_inline)) uint8_t low_walk(oracle_lowlevel *st, const int64_t *table, size_t table_len, const int16_t *padded_in,
                        size_t *frames_left, oracle_output_cb emit, const void *user, int norm_mode, uint64_t legacy_gain)
{
	/* clownresampler.h:1060-1091 */
	while (st->pos_int < *frames_left)
	{
		int64_t accum[ORACLE_MAX_CHANNELS] = {0};                   /* :1071 */

		frame_core(&st->cfg, table, table_len, accum, st->channels, padded_in, st->pos_int, st->pos_frac, norm_mode, legacy_gain);

		/* the position moves on BEFORE the frame is handed out (:1076-1078) */
		st->pos_frac += st->increment;
		st->pos_int += st->pos_frac / ORACLE_FRAC_ONE;
		st->pos_frac %= ORACLE_FRAC_ONE;

		if (!emit((void *)user, accum, st->channels))
		{
			/* consumer is full: give back the whole frames already stepped over (:1084-1088) */
			const uint64_t eaten = st->pos_int < *frames_left ? st->pos_int : *frames_left;

			*frames_left -= eaten;
			st->pos_int -= eaten;
			return 0;
		}
	}

	/* input exhausted: keep only the overshoot into the next chunk (:1065-1067) */
	st->pos_int -= *frames_left;
	*frames_left = 0;
	return 1;
}

uint8_t oracle_low_resample(oracle_lowlevel *st, const int64_t *table, size_t table_len, const int16_t *padded_in,
                            size_t *frames_left, oracle_output_cb emit, const void *user)
{
	return low_walk(st, table, table_len, padded_in, frames_left, emit, user, ORACLE_NORM_CURRENT, 0);
}

/* ------------------------------------------------------------------------- */
/* High-level API (clownresampler.h:1101-1176, :1183-1209, :1216-1250)       */
/* ------------------------------------------------------------------------- */

uint8_t oracle_high_init(oracle_highlevel *st, unsigned radius, uint32_t channels, uint64_t in_rate, uint64_t out_rate, uint64_t lowpass_rate)
{
	if (channels > ORACLE_MAX_CHANNELS)                             /* :1103 */
		return 0;

	if (!oracle_low_init(&st->low, radius, channels, in_rate, out_rate, lowpass_rate))
		return 0;

	/* :1109 - all three counters start at the radius in frames */
	st->max_radius_frames = st->lead_needed = st->trail_left = st->low.cfg.radius_frames;

	/* :1112-1115 - silent left halo; the window is empty and sits just after it */
	memset(st->staging, 0, st->max_radius_frames * channels * sizeof(st->staging[0]));
	st->win_begin = st->win_end = st->staging + st->max_radius_frames * channels;

	return 1;
}

uint8_t oracle_high_resample(oracle_highlevel *st, const int64_t *table, size_t table_len, oracle_input_cb pull,
                             oracle_output_cb emit, const void *user)
{
	const size_t ch = st->low.channels;
	const size_t halo = st->max_radius_frames * ch;                 /* samples, :1124 */
	uint8_t consumer_full = 0;

	/* :1127-1136 - first fill the look-ahead halo, right-aligned in staging[halo .. 2*halo) */
	while (st->lead_needed != 0)
	{
		const size_t got = pull((void *)user, st->staging + 2 * halo - st->lead_needed * ch, st->lead_needed);

		if (got == 0)
			return 1;

		st->lead_needed -= got;
	}

	do
	{
		if (st->win_begin == st->win_end)
		{
			/* :1150-1158 - slide the last 2*halo samples (left halo + look-ahead) to the front,
			   then pull fresh frames in behind them */
			memmove(st->staging, st->win_end - halo, 2 * halo * sizeof(st->staging[0]));
			st->win_begin = st->staging + halo;
			st->win_end = st->win_begin + pull((void *)user, st->staging + 2 * halo, (ORACLE_STAGING_SAMPLES - 2 * halo) / ch) * ch;

			if (st->win_begin == st->win_end)
				return 1;
		}

		{
			/* :1163-1171 - the low-level call sees the window flanked by the CURRENT radius */
			const size_t cur_halo = st->low.cfg.radius_frames * ch;
			size_t frames = (size_t)(st->win_end - st->win_begin) / ch;

			consumer_full = !oracle_low_resample(&st->low, table, table_len, st->win_begin - cur_halo, &frames, emit, user);
			st->win_begin = st->win_end - frames * ch;
		}
	} while (!consumer_full);

	return 0;
}

uint8_t oracle_high_adjust(oracle_highlevel *st, unsigned radius, uint64_t in_rate, uint64_t out_rate, uint64_t lowpass_rate)
{
	const oracle_lowlevel saved = st->low;                          /* :1186 */

	/* :1188-1206 - three ways to fail, each restores the saved state */
	if (!oracle_low_adjust(&st->low, radius, in_rate, out_rate, lowpass_rate)
	 || st->low.cfg.radius_frames > st->max_radius_frames
	 || st->low.cfg.radius_frames * 2 >= ORACLE_STAGING_SAMPLES / st->low.channels)
	{
		st->low = saved;
		return 0;
	}

	return 1;
}

typedef struct flush_ctx
{
	oracle_highlevel *st;
	oracle_output_cb emit;
	void *user;
} flush_ctx;

/* :1223-1233 - feeds the remaining trailing silence */
static size_t flush_pull(void *user, int16_t *buffer, size_t max_frames)
{
	flush_ctx *ctx = (flush_ctx *)user;
	const size_t n = max_frames < ctx->st->trail_left ? max_frames : ctx->st->trail_left;

	memset(buffer, 0, n * ctx->st->low.channels * sizeof(*buffer));
	ctx->st->trail_left -= n;
	return n;
}

/* :1235-1240 */
static uint8_t flush_emit(void *user, const int64_t *frame, uint32_t samples)
{
	flush_ctx *ctx = (flush_ctx *)user;
	return ctx->emit(ctx->user, frame, samples);
}

uint8_t oracle_high_end(oracle_highlevel *st, const int64_t *table, size_t table_len, oracle_output_cb emit, const void *user)
{
	/* :1242-1250 */
	flush_ctx ctx;
	ctx.st = st;
	ctx.emit = emit;
	ctx.user = (void *)user;
	return oracle_high_resample(st, table, table_len, flush_pull, flush_emit, &ctx);
}

/* ------------------------------------------------------------------------- */
/* Harness conveniences                                                      */
/* ------------------------------------------------------------------------- */

uint64_t oracle_count_output_frames(const oracle_lowlevel *st, uint64_t frames)
{
	/* frames are emitted while pos_int < frames, i.e. while P0 + j*inc < frames*65536
	   (verified against the reference by tests/test_oracle_vs_ref.py) */
	const unsigned __int128 start = (unsigned __int128)st->pos_int * ORACLE_FRAC_ONE + st->pos_frac;
	const unsigned __int128 limit = (unsigned __int128)frames * ORACLE_FRAC_ONE;

	if (start >= limit)
		return 0;

	return (uint64_t)((limit - start + st->increment - 1) / st->increment);
}

typedef struct store_ctx
{
	int32_t *out;
	size_t written;   /* frames */
	size_t capacity;  /* frames */
} store_ctx;

/* the reference harness's callback writes 4 little-endian bytes per sample (tests/test-low-level.c:43-49) */
static uint8_t store_emit(void *user, const int64_t *frame, uint32_t samples)
{
	store_ctx *ctx = (store_ctx *)user;
	int32_t *dst = ctx->out + ctx->written * samples;
	uint32_t c;

	for (c = 0; c < samples; ++c)
		dst[c] = (int32_t)(uint32_t)(uint64_t)frame[c];

	return ++ctx->written < ctx->capacity;
}

size_t oracle_low_resample_i32(oracle_lowlevel *st, const int64_t *table, size_t table_len, const int16_t *padded_in,
                               size_t *frames_left, int32_t *out, size_t out_capacity_frames, int norm_mode,
                               uint64_t legacy_gain, uint8_t *ran_out_of_input)
{
	store_ctx ctx;
	uint8_t exhausted;

	ctx.out = out;
	ctx.written = 0;
	ctx.capacity = out_capacity_frames;

	/* no room at all: nothing can be emitted, so only the "input already exhausted" exit may run */
	if (out_capacity_frames == 0 && st->pos_int < *frames_left)
	{
		if (ran_out_of_input != NULL)
			*ran_out_of_input = 0;
		return 0;
	}

	exhausted = low_walk(st, table, table_len, padded_in, frames_left, store_emit, &ctx, norm_mode, legacy_gain);

	if (ran_out_of_input != NULL)
		*ran_out_of_input = exhausted;

	return ctx.written;
}

typedef struct pcm_source
{
	const int16_t *pcm;
	size_t frames_left;
	size_t chunk;
	uint32_t channels;
	store_ctx sink;
} pcm_source;

static size_t pcm_pull(void *user, int16_t *buffer, size_t max_frames)
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

static uint8_t pcm_emit(void *user, const int64_t *frame, uint32_t samples)
{
	pcm_source *src = (pcm_source *)user;
	return store_emit(&src->sink, frame, samples);
}

size_t oracle_high_run_i32(oracle_highlevel *st, const int64_t *table, size_t table_len, const int16_t *pcm,
                           size_t pcm_frames, size_t pull_chunk, int32_t *out, size_t out_capacity_frames)
{
	pcm_source src;

	src.pcm = pcm;
	src.frames_left = pcm_frames;
	src.chunk = pull_chunk;
	src.channels = st->low.channels;
	src.sink.out = out;
	src.sink.written = 0;
	src.sink.capacity = out_capacity_frames;

	/* tests/test-high-level.c:126-127: Resample until the input dries up, then ResampleEnd */
	if (oracle_high_resample(st, table, table_len, pcm_pull, pcm_emit, &src))
		oracle_high_end(st, table, table_len, pcm_emit, &src);

	return src.sink.written;
}

uint64_t oracle_stream_hash(const int32_t *samples, size_t count, uint64_t seed)
{
	uint64_t h = seed;
	size_t i;

	for (i = 0; i < count; ++i)
		h = (h ^ (uint64_t)(uint32_t)samples[i]) * 1099511628211ull;

	return h;
}

uint64_t oracle_fill_noise(int16_t *dst, size_t samples, uint64_t state)
{
	size_t i;

	for (i = 0; i < samples; ++i)
	{
		state ^= state << 13;
		state ^= state >> 7;
		state ^= state << 17;
		dst[i] = (int16_t)(state >> 48);
	}

	return state;
}

/* ------------------------------------------------------------------------- */
/* Multi-threaded baseline driver                                            */
/* ------------------------------------------------------------------------- */

typedef struct mt_job
{
	oracle_lowlevel st;
	const int64_t *table;
	size_t table_len;
	const int16_t *in;      /* padded pointer for this range */
	size_t frames;          /* input frames in this range */
	int32_t *out;
	size_t capacity;
	size_t written;
} mt_job;

static void *mt_worker(void *arg)
{
	mt_job *job = (mt_job *)arg;
	size_t left = job->frames;

	job->written = oracle_low_resample_i32(&job->st, job->table, job->table_len, job->in, &left, job->out, job->capacity,
	                                       ORACLE_NORM_CURRENT, 0, NULL);
	return NULL;
}

size_t oracle_low_resample_i32_mt(const oracle_lowlevel *fresh, const int64_t *table, size_t table_len,
                                  const int16_t *padded_in, size_t frames, int32_t *out, unsigned threads)
{
	/* contiguous input-frame ranges [a,b); start state from the closed form of SURVEY.md 8(d):
	   k0 = ceil(a*65536/inc), P = k0*inc, pos_int = (P >> 16) - a, pos_frac = P & 0xFFFF,
	   pointer advanced by a frames (legal per clownresampler.h:725-733: the halo is real neighbours) */
	mt_job *jobs;
	pthread_t *tids;
	size_t total = 0;
	unsigned t;

	if (threads == 0)
		threads = 1;

	jobs = (mt_job *)calloc(threads, sizeof(*jobs));
	tids = (pthread_t *)calloc(threads, sizeof(*tids));

	for (t = 0; t < threads; ++t)
	{
		const uint64_t a = (uint64_t)frames * t / threads;
		const uint64_t b = (uint64_t)frames * (t + 1) / threads;
		/* (a state that is not fresh - a shard of a longer stream, a resumed call - starts its timeline at p0 instead of 0) */
		const unsigned __int128 p0 = (unsigned __int128)fresh->pos_int * ORACLE_FRAC_ONE + fresh->pos_frac;
		const unsigned __int128 a_pos = (unsigned __int128)a * ORACLE_FRAC_ONE;
		const unsigned __int128 first_k = a_pos > p0 ? (a_pos - p0 + fresh->increment - 1) / fresh->increment : 0;
		const unsigned __int128 first_p = p0 + first_k * fresh->increment;

		jobs[t].st = *fresh;
		jobs[t].st.pos_int = (uint64_t)(first_p / ORACLE_FRAC_ONE) - a;
		jobs[t].st.pos_frac = (uint64_t)(first_p % ORACLE_FRAC_ONE);
		jobs[t].table = table;
		jobs[t].table_len = table_len;
		jobs[t].in = padded_in + a * fresh->channels;
		jobs[t].frames = (size_t)(b - a);
		jobs[t].out = out + (size_t)first_k * fresh->channels;
		jobs[t].capacity = (size_t)-1;
	}

	for (t = 1; t < threads; ++t)
		pthread_create(&tids[t], NULL, mt_worker, &jobs[t]);

	mt_worker(&jobs[0]);

	for (t = 1; t < threads; ++t)
		pthread_join(tids[t], NULL);

	for (t = 0; t < threads; ++t)
		total += jobs[t].written;

	free(jobs);
	free(tids);
	return total;
}
