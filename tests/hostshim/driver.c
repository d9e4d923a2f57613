/*
 * driver.c - TEST INFRASTRUCTURE: drives the product's HOST logic (the C sources of clownresampler_amd/csrc built against tests/hostshim/crhip_fake.c,
 * see the Makefile) through its C ABI and holds every result to the oracle (oracle/cr_oracle.c, linked into the same test library).
 * Built three times: plain, -fsanitize=address,undefined and -fsanitize=thread; tests/test_hostshim.py runs them.
 *
 *   driver [test ...]     no argument: every test.  Exit code 0 = all passed.
 *
 * What it reaches: the bulk entry point incl. its 4 Mi-frame batching and host pipeline thread, the callback API's
 * early stop / resume and compute-ahead thread, the high-level streaming API with Adjust at any streaming window, concurrent callers over
 * a small plan cache (LRU eviction under load), variable-rate segments and their validation, the sharded call on several (fake) devices,
 * device-resident launches from many threads, Shutdown and use after it.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "clownresampler_amd.h"
#include "cr_oracle.h"

static ClownResampler_Precomputed pre;
static int64_t *table;
static size_t table_len;
static int failures = 0;

#define CHECK(cond, ...)                                                                                                           \
	do                                                                                                                             \
	{                                                                                                                              \
		if (!(cond))                                                                                                               \
		{                                                                                                                          \
			fprintf(stderr, "FAIL %s:%d: ", __func__, __LINE__);                                                                   \
			fprintf(stderr, __VA_ARGS__);                                                                                          \
			fprintf(stderr, "\n");                                                                                                 \
			__atomic_fetch_add(&failures, 1, __ATOMIC_RELAXED);                                                                    \
			return;                                                                                                                \
		}                                                                                                                          \
	} while (0)

static void on_error(int code, const char *message, void *user)
{
	(void)user;
	fprintf(stderr, "library error %d: %s\n", code, message);
}

static int16_t *noise(size_t samples, uint64_t seed)
{
	int16_t *p = (int16_t *)malloc((samples != 0 ? samples : 1) * sizeof(int16_t));
	oracle_fill_noise(p, samples, seed);
	return p;
}

/* zero halo of `radius` frames either side (tests/test-low-level.c:133-152) */
static int16_t *padded_noise(size_t frames, unsigned channels, size_t radius, uint64_t seed)
{
	int16_t *p = (int16_t *)calloc((frames + 2 * radius) * channels + 1, sizeof(int16_t));
	oracle_fill_noise(p + radius * channels, frames * channels, seed);
	return p;
}

static int states_equal(const ClownResampler_LowLevel_State *a, const oracle_lowlevel *b)
{
	return a->lowest_level.stretched_kernel_radius == b->cfg.stretched_radius && a->lowest_level.integer_stretched_kernel_radius == b->cfg.radius_frames
	    && a->lowest_level.stretched_kernel_radius_delta == b->cfg.radius_delta && a->lowest_level.kernel_step_size == b->cfg.table_step
	    && a->channels == b->channels && a->position_integer == b->pos_int && a->position_fractional == b->pos_frac && a->increment == b->increment;
}

/* ---- 1. the bulk entry point: one shot, chunks with carried state, capacity stops ---- */
static void one_bulk_case(unsigned channels, unsigned long in_rate, unsigned long out_rate, unsigned long low_pass, size_t frames, size_t chunk, size_t capacity)
{
	ClownResampler_LowLevel_State st;
	oracle_lowlevel ost;
	const int ok_a = ClownResampler_LowLevel_Init(&st, channels, in_rate, out_rate, low_pass), ok_b = oracle_low_init(&ost, 3, channels, in_rate, out_rate, low_pass);
	size_t radius, total, pos = 0, w = 0, w2 = 0, guard = 0;
	int16_t *pcm;
	int32_t *got, *want;

	CHECK(ok_a == ok_b && ok_a, "init %u %lu %lu %lu", channels, in_rate, out_rate, low_pass);
	radius = ost.cfg.radius_frames;
	pcm = padded_noise(frames, channels, radius, 1000 + frames + channels);
	total = oracle_count_output_frames(&ost, frames);
	got = (int32_t *)malloc((total + 2) * channels * sizeof(int32_t));
	want = (int32_t *)malloc((total + 2) * channels * sizeof(int32_t));
	while (pos < frames)
	{
		/* a chunk of the input with the state carried (the halo of a middle chunk is its real neighbours: clownresampler.h:725-733),
		   the output possibly too small for it (the caller comes back with what is left: examples/low-level.c:87-102) */
		size_t n = chunk != 0 && chunk < frames - pos ? chunk : frames - pos, left_a = n, left_b = n;
		cc_bool ran_a = 0;
		uint8_t ran_b = 0;
		const size_t cap = capacity != 0 ? capacity : total + 1;
		const size_t a = ClownResampler_LowLevel_ResampleBulk(&st, &pre, pcm + pos * channels, &left_a, got + w * channels, cap < total + 1 - w ? cap : total + 1 - w, &ran_a);
		const size_t b = oracle_low_resample_i32(&ost, table, table_len, pcm + pos * channels, &left_b, want + w2 * channels, cap < total + 1 - w2 ? cap : total + 1 - w2, ORACLE_NORM_CURRENT, 0, &ran_b);

		CHECK(a == b && left_a == left_b && (ran_a != 0) == (ran_b != 0) && states_equal(&st, &ost), "bulk step: %zu/%zu frames, %zu/%zu left, ran out %d/%d", a, b, left_a, left_b, ran_a, ran_b);
		w += a;
		w2 += b;
		pos += n - left_a;
		CHECK(++guard < 100000, "no progress");
	}
	CHECK(w == total && memcmp(got, want, total * channels * sizeof(int32_t)) == 0, "bulk %u ch %lu -> %lu: samples differ", channels, in_rate, out_rate);
	free(pcm);
	free(got);
	free(want);
}

static void test_bulk(void)
{
	one_bulk_case(2, 44100, 48000, 44100, 300000, 0, 0);
	one_bulk_case(1, 48000, 44100, 44100, 123457, 0, 0);
	one_bulk_case(8, 48000, 44100, 44100, 50000, 0, 0);
	one_bulk_case(3, 48000, 11025, 11025, 80000, 4099, 0);
	one_bulk_case(16, 44100, 48000, 44100, 20000, 1, 0);
	one_bulk_case(2, 44100, 8000, 8000, 200000, 0, 777);
	one_bulk_case(5, 8000, 96000, 8000, 9000, 1234, 1000);
	one_bulk_case(2, 44100, 48000, 44100, 0, 0, 0);
	one_bulk_case(2, 44100, 1000, 1000, 30000, 0, 0);   /* a long window (generic kernel territory) */
	one_bulk_case(2, 1, 1, 1, 5000, 17, 0);
	printf("ok bulk\n");
}

/* ---- 2. a call beyond the 4 Mi-frame batches: the host pipeline thread ---- */
static void test_bulk_batches(void)
{
	const unsigned channels = 1;
	const size_t frames = 9500000;   /* 10.3 M output frames: three batches */
	ClownResampler_LowLevel_State st;
	oracle_lowlevel ost;
	int16_t *pcm;
	int32_t *got, *want;
	size_t total, left = frames, n, m;
	cc_bool ran_out = 0;

	ClownResampler_LowLevel_Init(&st, channels, 44100, 48000, 44100);
	oracle_low_init(&ost, 3, channels, 44100, 48000, 44100);
	pcm = padded_noise(frames, channels, ost.cfg.radius_frames, 99);
	total = oracle_count_output_frames(&ost, frames);
	got = (int32_t *)malloc((total + 1) * sizeof(int32_t));
	want = (int32_t *)malloc((total + 1) * sizeof(int32_t));
	n = ClownResampler_LowLevel_ResampleBulk(&st, &pre, pcm, &left, got, total + 1, &ran_out);
	m = oracle_low_resample_i32_mt(&ost, table, table_len, pcm, frames, want, 8);
	CHECK(n == total && m == total && left == 0 && ran_out, "batched bulk: %zu of %zu frames, %zu left", n, total, left);
	CHECK(memcmp(got, want, total * sizeof(int32_t)) == 0, "batched bulk: samples differ");
	free(pcm);
	free(got);
	free(want);
	printf("ok bulk_batches\n");
}

/* ---- 3. the reference's callback signature: early stops, resume, the compute-ahead thread ---- */
typedef struct sink
{
	int32_t *out;
	size_t at, stop_at;   /* samples written; the callback returns 0 on the frame that reaches stop_at samples */
} sink;

static cc_bool sink_frame(void *user, const cc_s32f *frame, cc_u8f samples)
{
	sink *s = (sink *)user;
	cc_u8f i;

	for (i = 0; i < samples; ++i)
		s->out[s->at++] = (int32_t)frame[i];
	return s->at < s->stop_at;
}

static uint8_t sink_frame_oracle(void *user, const int64_t *frame, uint32_t samples)
{
	sink *s = (sink *)user;
	uint32_t i;

	for (i = 0; i < samples; ++i)
		s->out[s->at++] = (int32_t)frame[i];
	return s->at < s->stop_at;
}

static void one_callback_case(unsigned channels, size_t frames, size_t stop_every)
{
	ClownResampler_LowLevel_State st;
	oracle_lowlevel ost;
	int16_t *pcm;
	sink a, b;
	size_t total, pos = 0, left_a = frames, left_b = frames, guard = 0;

	ClownResampler_LowLevel_Init(&st, channels, 44100, 48000, 44100);
	oracle_low_init(&ost, 3, channels, 44100, 48000, 44100);
	pcm = padded_noise(frames, channels, ost.cfg.radius_frames, 7 + frames);
	total = oracle_count_output_frames(&ost, frames);
	a.out = (int32_t *)malloc((total + 1) * channels * sizeof(int32_t));
	b.out = (int32_t *)malloc((total + 1) * channels * sizeof(int32_t));
	a.at = b.at = 0;
	for (;;)
	{
		const size_t before = left_a;
		cc_bool ra;
		uint8_t rb;

		a.stop_at = b.stop_at = stop_every != 0 ? a.at + stop_every * channels : (size_t)-1;
		ra = ClownResampler_LowLevel_Resample(&st, &pre, pcm + pos * channels, &left_a, sink_frame, &a);
		rb = oracle_low_resample(&ost, table, table_len, pcm + pos * channels, &left_b, sink_frame_oracle, &b);
		CHECK((ra != 0) == (rb != 0) && left_a == left_b && a.at == b.at && states_equal(&st, &ost), "callback step: returns %d/%d, left %zu/%zu, samples %zu/%zu", ra, rb, left_a, left_b, a.at, b.at);
		pos += before - left_a;
		if (ra)
			break;
		CHECK(++guard < 1000000, "no progress");
	}
	CHECK(a.at == total * channels && memcmp(a.out, b.out, a.at * sizeof(int32_t)) == 0, "callback: samples differ");
	free(pcm);
	free(a.out);
	free(b.out);
}

static void test_callback(void)
{
	one_callback_case(2, 60000, 1000);
	one_callback_case(1, 50000, 777);
	one_callback_case(3, 20000, 0);
	one_callback_case(2, 3200000, 0);         /* past the growing batches: the helper thread computes a batch ahead */
	one_callback_case(2, 3200000, 1300000);   /* ... and is stopped in the pipelined part, twice */
	printf("ok callback\n");
}

/* ---- 4. the streaming API with Adjust, at three streaming windows ---- */
typedef struct source
{
	const int16_t *pcm;
	size_t frames, at, chunk;
	unsigned channels;
} source;

static size_t pull(void *user, cc_s16l *buffer, size_t max_frames)
{
	source *s = (source *)user;
	size_t n = s->frames - s->at < max_frames ? s->frames - s->at : max_frames;

	if (s->chunk != 0 && n > s->chunk)
		n = s->chunk;
	memcpy(buffer, s->pcm + s->at * s->channels, n * s->channels * sizeof(int16_t));
	s->at += n;
	return n;
}

typedef struct both
{
	source src;
	sink out;
} both;

static size_t pull_both(void *user, cc_s16l *buffer, size_t max_frames) { return pull(&((both *)user)->src, buffer, max_frames); }
static cc_bool sink_both(void *user, const cc_s32f *frame, cc_u8f samples) { return sink_frame(&((both *)user)->out, frame, samples); }
static size_t pull_both_oracle(void *user, int16_t *buffer, size_t max_frames) { return pull(&((both *)user)->src, buffer, max_frames); }
static uint8_t sink_both_oracle(void *user, const int64_t *frame, uint32_t samples) { return sink_frame_oracle(&((both *)user)->out, frame, samples); }

static void one_stream_case(size_t window, unsigned channels, size_t frames, size_t chunk, uint64_t seed)
{
	static ClownResampler_HighLevel_State hs;   /* (static: 8 KB, and the side window is keyed by its address) */
	oracle_highlevel ohs;
	both a, b;
	int16_t *pcm = noise(frames * channels, seed);
	const size_t room = (frames * 13 + 4096) * channels;
	/* triples applied between calls: accepted, a narrower kernel, a wider one (rejected), a zero rate (rejected) */
	static const unsigned long triples[][3] = {{48000, 16000, 16000}, {48000, 24000, 24000}, {44100, 48000, 44100}, {48000, 8000, 4000}, {0, 48000, 48000}, {32000, 96000, 32000}, {48000, 16000, 16000}};
	uint64_t rng = seed * 2654435761u + 1;
	int step, ended = 0, dry = 0;

	ClownResamplerAMD_SetStreamingWindow(window);
	CHECK(ClownResampler_HighLevel_Init(&hs, channels, 48000, 16000, 16000) && oracle_high_init(&ohs, 3, channels, 48000, 16000, 16000), "init");
	memset(&a, 0, sizeof(a));
	memset(&b, 0, sizeof(b));
	a.src.pcm = b.src.pcm = pcm;
	a.src.frames = b.src.frames = frames;
	a.src.chunk = b.src.chunk = chunk;
	a.src.channels = b.src.channels = channels;
	a.out.out = (int32_t *)malloc(room * sizeof(int32_t));
	b.out.out = (int32_t *)malloc(room * sizeof(int32_t));
	for (step = 0; step < 400 && ended < 3; ++step)
	{
		cc_bool ra;
		uint8_t rb;
		size_t budget;

		rng = rng * 6364136223846793005ull + 1442695040888963407ull;
		budget = (size_t)((rng >> 33) % 3000) + 1;
		a.out.stop_at = a.out.at + budget * channels;
		b.out.stop_at = b.out.at + budget * channels;
		CHECK(a.out.stop_at < room, "output room");
		if (!dry)
		{
			ra = ClownResampler_HighLevel_Resample(&hs, &pre, pull_both, sink_both, &a);
			rb = oracle_high_resample(&ohs, table, table_len, pull_both_oracle, sink_both_oracle, &b);
			dry = ra != 0;
		}
		else
		{
			ra = ClownResampler_HighLevel_ResampleEnd(&hs, &pre, sink_both, &a);
			rb = oracle_high_end(&ohs, table, table_len, sink_both_oracle, &b);
			ended += ra != 0;
		}
		CHECK((ra != 0) == (rb != 0) && a.out.at == b.out.at && states_equal(&hs.low_level, &ohs.low) && hs.leading_padding_frames_needed == ohs.lead_needed
		          && hs.trailing_padding_frames_remaining == ohs.trail_left,
		      "window %zu step %d: returns %d/%d, samples %zu/%zu, position %zu/%llu", window, step, ra, rb, a.out.at, b.out.at, hs.low_level.position_integer, (unsigned long long)ohs.low.pos_int);
		if ((rng >> 20) % 3 == 0)
		{
			const unsigned long *t = triples[(rng >> 40) % (sizeof(triples) / sizeof(triples[0]))];
			const cc_bool ka = ClownResampler_HighLevel_Adjust(&hs, t[0], t[1], t[2]);
			const uint8_t kb = oracle_high_adjust(&ohs, 3, t[0], t[1], t[2]);
			CHECK((ka != 0) == (kb != 0) && states_equal(&hs.low_level, &ohs.low), "Adjust %lu %lu %lu: %d/%d", t[0], t[1], t[2], ka, kb);
		}
	}
	CHECK(ended >= 1 && a.out.at == b.out.at && memcmp(a.out.out, b.out.out, a.out.at * sizeof(int32_t)) == 0, "stream (window %zu): samples differ", window);
	ClownResamplerAMD_HighLevel_Release(&hs);
	free(pcm);
	free(a.out.out);
	free(b.out.out);
}

static void test_stream(void)
{
	static const size_t windows[3] = {0, 5000, (size_t)1 << 18};
	unsigned w;

	for (w = 0; w < 3; ++w)
	{
		one_stream_case(windows[w], 2, 40000, 333, 11 + w);
		one_stream_case(windows[w], 5, 9000, 0, 21 + w);
		one_stream_case(windows[w], 1, 150000, 100000, 31 + w);
	}
	ClownResamplerAMD_SetStreamingWindow((size_t)1 << 18);
	printf("ok stream\n");
}

/* ---- 5. concurrent callers over a plan cache of three ---- */
static void *caller(void *argument)
{
	const unsigned id = (unsigned)(size_t)argument;
	static const unsigned long rates[][3] = {{44100, 48000, 44100}, {48000, 44100, 44100}, {44100, 8000, 8000}, {8000, 44100, 8000}, {32000, 48000, 32000}, {48000, 32000, 32000}, {22050, 44100, 22050}};
	unsigned k;

	for (k = 0; k < 14; ++k)
	{
		const unsigned long *r = rates[(id + k) % 7];
		one_bulk_case(1 + (id + k) % 4, r[0], r[1], r[2], 20000 + 1000 * id + 37 * k, k % 3 == 0 ? 5000 : 0, 0);
	}
	return NULL;
}

static void test_concurrent(void)
{
	pthread_t threads[6];
	unsigned i;

	ClownResamplerAMD_SetPlanCacheLimit(3);
	for (i = 0; i < 6; ++i)
		pthread_create(&threads[i], NULL, caller, (void *)(size_t)i);
	for (i = 0; i < 6; ++i)
		pthread_join(threads[i], NULL);
	CHECK(ClownResamplerAMD_PlanCacheCount() <= 3 + 6, "plan cache holds %zu plans", ClownResamplerAMD_PlanCacheCount());
	ClownResamplerAMD_SetPlanCacheLimit(64);
	printf("ok concurrent\n");
}

/* ---- 6. variable rate on the "device": segments, and what must be refused before anything is launched ---- */
static void test_segments(void)
{
	const unsigned channels = 2;
	const size_t halo = 24, frames = 60000;
	ClownResamplerAMD_Segment segments[5] = {{20000, 44100, 48000, 44100}, {5000, 48000, 44100, 44100}, {1, 44100, 44100, 22050}, {14999, 44100, 88200, 44100}, {20000, 44100, 22050, 22050}};
	size_t counts[5], n, i, done = 0, pos = 0;
	ClownResampler_LowLevel_State st;
	oracle_lowlevel ost;
	int16_t *pcm = padded_noise(frames, channels, halo, 5);
	int16_t *d_in = (int16_t *)ClownResamplerAMD_DeviceAlloc((frames + 2 * halo) * channels * sizeof(int16_t));
	const size_t room = frames * 3;
	int32_t *d_out = (int32_t *)ClownResamplerAMD_DeviceAlloc(room * channels * sizeof(int32_t));
	int32_t *want = (int32_t *)malloc(room * channels * sizeof(int32_t));
	int mode;

	ClownResamplerAMD_CopyToDevice(d_in, pcm, (frames + 2 * halo) * channels * sizeof(int16_t));
	for (mode = 0; mode < 3; ++mode)
	{
		ClownResamplerAMD_DebugSegmentsMode(mode);
		ClownResampler_LowLevel_Init(&st, channels, 44100, 48000, 44100);
		oracle_low_init(&ost, 3, channels, 44100, 48000, 44100);
		n = ClownResamplerAMD_ResampleSegmentsDevice(&st, &pre, d_in + halo * channels, halo, segments, 5, d_out, room, 0, counts, NULL);
		ClownResamplerAMD_StreamSynchronize(NULL);
		done = 0;
		pos = 0;
		for (i = 0; i < 5; ++i)
		{
			size_t left = segments[i].input_frames, m;
			uint8_t ran = 0;

			oracle_low_adjust(&ost, 3, segments[i].input_sample_rate, segments[i].output_sample_rate, segments[i].low_pass_filter_sample_rate);
			m = oracle_low_resample_i32(&ost, table, table_len, pcm + (halo + pos - ost.cfg.radius_frames) * channels, &left, want + done * channels, room - done, ORACLE_NORM_CURRENT, 0, &ran);
			CHECK(m == counts[i] && left == 0, "segments mode %d: segment %zu has %zu frames, the oracle %zu", mode, i, counts[i], m);
			done += m;
			pos += segments[i].input_frames;
		}
		CHECK(n == done && states_equal(&st, &ost) && memcmp(d_out, want, done * channels * sizeof(int32_t)) == 0, "segments mode %d: %zu/%zu frames or samples differ", mode, n, done);
	}
	ClownResamplerAMD_DebugSegmentsMode(0);
	/* refused before anything is launched: a rejected triple, a halo too small, an output too small (state untouched, 0 returned) */
	{
		ClownResamplerAMD_Segment bad[2] = {{1000, 44100, 48000, 44100}, {1000, 0, 48000, 48000}};
		ClownResampler_LowLevel_State before;
		const unsigned long long launches = ClownResamplerAMD_DebugLaunchCount(0) + ClownResamplerAMD_DebugLaunchCount(1);

		ClownResamplerAMD_SetErrorHandler(NULL == NULL ? on_error : NULL, NULL);
		ClownResampler_LowLevel_Init(&st, channels, 44100, 48000, 44100);
		before = st;
		CHECK(ClownResamplerAMD_ResampleSegmentsDevice(&st, &pre, d_in + halo * channels, halo, bad, 2, d_out, room, 0, NULL, NULL) == 0 && ClownResamplerAMD_LastErrorCode() != 0, "a zero rate was accepted");
		ClownResamplerAMD_ClearError();
		bad[1].input_sample_rate = 44100;
		bad[1].low_pass_filter_sample_rate = 1000;   /* 133 frames of radius: more than the halo */
		CHECK(ClownResamplerAMD_ResampleSegmentsDevice(&st, &pre, d_in + halo * channels, halo, bad, 2, d_out, room, 0, NULL, NULL) == 0 && ClownResamplerAMD_LastErrorCode() != 0, "a halo too small was accepted");
		ClownResamplerAMD_ClearError();
		CHECK(ClownResamplerAMD_ResampleSegmentsDevice(&st, &pre, d_in + halo * channels, halo, segments, 5, d_out, 100, 0, NULL, NULL) == 0 && ClownResamplerAMD_LastErrorCode() != 0, "an output too small was accepted");
		ClownResamplerAMD_ClearError();
		CHECK(memcmp(&st, &before, sizeof(st)) == 0 && ClownResamplerAMD_DebugLaunchCount(0) + ClownResamplerAMD_DebugLaunchCount(1) == launches, "a refused call touched the state or launched");
	}
	ClownResamplerAMD_DeviceFree(d_in);
	ClownResamplerAMD_DeviceFree(d_out);
	free(pcm);
	free(want);
	printf("ok segments\n");
}

/* ---- 7. one stream over several devices, one call (eight shards on the fake devices) ---- */
static void test_sharded(void)
{
	enum { SHARDS = 8 };
	const unsigned channels = 2;
	const size_t frames = 400000;
	const int devices = ClownResamplerAMD_DeviceCount();
	ClownResampler_LowLevel_State st, one;
	oracle_lowlevel ost;
	ClownResamplerAMD_DeviceShard args[SHARDS];
	ClownResamplerAMD_Shard plan[SHARDS];
	int16_t *pcm;
	int32_t *want, *root;
	size_t total, n, left = frames, radius;
	uint8_t ran = 0;
	unsigned r;

	ClownResampler_LowLevel_Init(&st, channels, 44100, 48000, 44100);
	one = st;
	oracle_low_init(&ost, 3, channels, 44100, 48000, 44100);
	radius = ost.cfg.radius_frames;
	pcm = padded_noise(frames, channels, radius, 4242);
	total = oracle_count_output_frames(&ost, frames);
	want = (int32_t *)malloc((total + 1) * channels * sizeof(int32_t));
	oracle_low_resample_i32(&ost, table, table_len, pcm, &left, want, total + 1, ORACLE_NORM_CURRENT, 0, &ran);
	for (r = 0; r < SHARDS; ++r)
	{
		size_t in_bytes, out_bytes;

		CHECK(ClownResamplerAMD_PlanShard(&one, frames, r, SHARDS, &plan[r]) == 0, "PlanShard %u", r);
		in_bytes = (plan[r].input_frames + 2 * plan[r].halo_frames) * channels * sizeof(int16_t);
		out_bytes = (plan[r].output_frames + 1) * channels * sizeof(int32_t);
		args[r].device = (int)(r % (unsigned)devices);
		args[r].device_input = ClownResamplerAMD_DeviceAllocOn(args[r].device, in_bytes);
		args[r].device_output = ClownResamplerAMD_DeviceAllocOn(args[r].device, out_bytes);
		args[r].hip_stream = NULL;
		ClownResamplerAMD_SetThreadDevice(args[r].device);
		ClownResamplerAMD_CopyToDevice((void *)args[r].device_input, pcm + plan[r].first_input_frame * channels, in_bytes);
	}
	ClownResamplerAMD_SetThreadDevice(-1);
	root = (int32_t *)ClownResamplerAMD_DeviceAllocOn(0, (total + 1) * channels * sizeof(int32_t));
	n = ClownResamplerAMD_ResampleShardedDevice(&st, &pre, frames, args, SHARDS, 0, CLOWNRESAMPLER_AMD_GATHER_PEER_COPY, 0, root);
	ClownResamplerAMD_ShardedSynchronize(args, SHARDS);
	CHECK(n == total && states_equal(&st, &ost), "sharded: %zu of %zu frames", n, total);
	CHECK(memcmp(root, want, total * channels * sizeof(int32_t)) == 0, "sharded: the concatenated stream differs");
	for (r = 0; r < SHARDS; ++r)
	{
		ClownResamplerAMD_DeviceFree((void *)args[r].device_input);
		ClownResamplerAMD_DeviceFree(args[r].device_output);
	}
	ClownResamplerAMD_DeviceFree(root);
	free(pcm);
	free(want);
	printf("ok sharded (%d fake devices)\n", devices);
}

/* ---- 8. device-resident launches of one plan from many threads ---- */
typedef struct launcher
{
	ClownResamplerAMD_Plan *plan;
	const ClownResampler_LowLevel_State *fresh;
	const int16_t *d_in;
	const int32_t *want;
	size_t frames, total;
	unsigned channels;
} launcher;

static void *launch_many(void *argument)
{
	const launcher *l = (const launcher *)argument;
	int32_t *d_out = (int32_t *)ClownResamplerAMD_DeviceAlloc((l->total + 1) * l->channels * sizeof(int32_t));
	unsigned k;

	for (k = 0; k < 10; ++k)
	{
		ClownResampler_LowLevel_State st = *l->fresh;
		size_t left = l->frames;
		cc_bool ran = 0;
		const size_t n = ClownResamplerAMD_ResampleDevice(l->plan, &st, l->d_in, &left, d_out, l->total + 1, NULL, &ran);

		ClownResamplerAMD_StreamSynchronize(NULL);
		if (n != l->total || left != 0 || !ran || memcmp(d_out, l->want, l->total * l->channels * sizeof(int32_t)) != 0)
		{
			fprintf(stderr, "FAIL launch_many: %zu of %zu frames or samples differ\n", n, l->total);
			__atomic_fetch_add(&failures, 1, __ATOMIC_RELAXED);
			break;
		}
	}
	ClownResamplerAMD_DeviceFree(d_out);
	return NULL;
}

static void test_device_threads(void)
{
	const unsigned channels = 2;
	const size_t frames = 70000;
	ClownResampler_LowLevel_State st;
	oracle_lowlevel ost;
	launcher l;
	pthread_t threads[5];
	int16_t *pcm, *d_in;
	int32_t *want;
	size_t left = frames, radius;
	uint8_t ran = 0;
	unsigned i;

	ClownResampler_LowLevel_Init(&st, channels, 48000, 44100, 44100);
	oracle_low_init(&ost, 3, channels, 48000, 44100, 44100);
	radius = ost.cfg.radius_frames;
	pcm = padded_noise(frames, channels, radius, 808);
	l.total = oracle_count_output_frames(&ost, frames);
	want = (int32_t *)malloc((l.total + 1) * channels * sizeof(int32_t));
	oracle_low_resample_i32(&ost, table, table_len, pcm, &left, want, l.total + 1, ORACLE_NORM_CURRENT, 0, &ran);
	d_in = (int16_t *)ClownResamplerAMD_DeviceAlloc((frames + 2 * radius) * channels * sizeof(int16_t));
	ClownResamplerAMD_CopyToDevice(d_in, pcm, (frames + 2 * radius) * channels * sizeof(int16_t));
	l.plan = ClownResamplerAMD_PlanCreate(&st, &pre);
	l.fresh = &st;
	l.d_in = d_in;
	l.want = want;
	l.frames = frames;
	l.channels = channels;
	CHECK(l.plan != NULL, "PlanCreate");
	for (i = 0; i < 5; ++i)
		pthread_create(&threads[i], NULL, launch_many, &l);
	for (i = 0; i < 5; ++i)
		pthread_join(threads[i], NULL);
	ClownResamplerAMD_DeviceFree(d_in);
	free(pcm);
	free(want);
	printf("ok device_threads\n");
}

/* ---- 9. Shutdown, and the library used again after it ---- */
static void test_shutdown(void)
{
	ClownResamplerAMD_Shutdown();
	CHECK(ClownResamplerAMD_PlanCacheCount() == 0 && ClownResamplerAMD_StreamingWindowCount() == 0, "Shutdown left %zu plans, %zu windows", ClownResamplerAMD_PlanCacheCount(), ClownResamplerAMD_StreamingWindowCount());
	one_bulk_case(2, 44100, 48000, 44100, 10000, 0, 0);
	ClownResamplerAMD_Shutdown();
	printf("ok shutdown\n");
}

/* ---- 10. a device that says no: every entry point hands the failure back - the reference's "callback said stop" outcome
   (clownresampler.h:746-748), the state and the input count as after what the consumer HAS been given - and the same call, repeated, is right ---- */
void crhip_fake_fail(int kind, int nth);
unsigned long long crhip_fake_failed(int kind);
enum { FAIL_MALLOC = 0, FAIL_HOST_ALLOC = 1, FAIL_LAUNCH = 2, FAIL_COPY = 3, FAIL_SYNC = 4, FAIL_KINDS = 5 };
static const char *const fail_names[FAIL_KINDS] = {"hipMalloc", "hipHostMalloc", "launch", "copy", "synchronise"};
static int quiet_errors = 0;

static void on_error_counted(int code, const char *message, void *user)
{
	(void)code;
	(void)message;
	++*(int *)user;
}

static void disarm(void)
{
	int k;

	for (k = 0; k < FAIL_KINDS; ++k)
		crhip_fake_fail(k, 0);
}

static void test_failures(void)
{
	const unsigned channels = 2;
	const size_t frames = 30000, radius = 3;
	int16_t *pcm = padded_noise(frames, channels, radius, 77);
	const size_t room = frames * 2;
	int32_t *out = (int32_t *)malloc(room * channels * sizeof(int32_t)), *want = (int32_t *)malloc(room * channels * sizeof(int32_t));
	int reported = 0, kind, nth;
	size_t want_frames;

	(void)quiet_errors;
	ClownResamplerAMD_SetErrorHandler(on_error_counted, &reported);
	{
		oracle_lowlevel ost;
		size_t left = frames;
		uint8_t ran = 0;

		oracle_low_init(&ost, 3, channels, 44100, 48000, 44100);
		want_frames = oracle_low_resample_i32(&ost, table, table_len, pcm, &left, want, room, ORACLE_NORM_CURRENT, 0, &ran);
	}

	for (kind = 0; kind < FAIL_KINDS; ++kind)
		for (nth = 1; nth <= 4; ++nth)
		{
			ClownResampler_LowLevel_State st, before;
			size_t left = frames, n;
			cc_bool ran_out = 7;
			unsigned long long failed_before = crhip_fake_failed(kind);
			int injected_here;

			/* every failure meets a COLD library: plans, staging sets and ticket rings all have to be made again */
			ClownResamplerAMD_Shutdown();
			ClownResamplerAMD_ClearError();
			reported = 0;

			/* (a) the bulk entry point */
			ClownResampler_LowLevel_Init(&st, channels, 44100, 48000, 44100);
			before = st;
			crhip_fake_fail(kind, nth);
			n = ClownResampler_LowLevel_ResampleBulk(&st, &pre, pcm, &left, out, room, &ran_out);
			disarm();
			injected_here = crhip_fake_failed(kind) != failed_before;
			if (injected_here)
			{
				CHECK(n == 0 && ran_out == cc_false && left == frames && memcmp(&st, &before, sizeof(st)) == 0 && ClownResamplerAMD_LastErrorCode() != 0 && reported != 0,
				      "bulk, %s #%d: %zu frames, ran_out %d, %zu left, error %d", fail_names[kind], nth, n, (int)ran_out, left, ClownResamplerAMD_LastErrorCode());
				ClownResamplerAMD_ClearError();
				n = ClownResampler_LowLevel_ResampleBulk(&st, &pre, pcm, &left, out, room, &ran_out);
			}
			CHECK(n == want_frames && left == 0 && ran_out == cc_true && memcmp(out, want, n * channels * sizeof(int32_t)) == 0 && ClownResamplerAMD_LastErrorCode() == 0,
			      "bulk after %s #%d: %zu of %zu frames", fail_names[kind], nth, n, want_frames);
		}

	/* (b) the callback form: a failure part-way = a stop after the frames already handed out; the rest on the next call */
	for (kind = 0; kind < FAIL_KINDS; ++kind)
		for (nth = 1; nth <= 12; nth += (nth < 4 ? 1 : 4))
		{
			ClownResampler_LowLevel_State st;
			oracle_lowlevel ost;
			sink got;
			size_t left = frames, calls = 0;
			unsigned long long failed_before = crhip_fake_failed(kind);
			cc_bool r = cc_false;

			ClownResamplerAMD_Shutdown();
			ClownResamplerAMD_ClearError();
			memset(&got, 0, sizeof(got));
			got.out = out;
			got.at = 0;
			got.stop_at = (size_t)-1;
			ClownResampler_LowLevel_Init(&st, channels, 44100, 48000, 44100);
			crhip_fake_fail(kind, nth);
			while (calls < 8)
			{
				const size_t consumed_before = frames - left;

				r = ClownResampler_LowLevel_Resample(&st, &pre, pcm + consumed_before * channels, &left, sink_frame, &got);
				disarm();
				++calls;
				if (r)
					break;
				CHECK(ClownResamplerAMD_LastErrorCode() != 0, "callback, %s #%d: cc_false without a stop or an error", fail_names[kind], nth);
				ClownResamplerAMD_ClearError();
			}
			oracle_low_init(&ost, 3, channels, 44100, 48000, 44100);
			(void)ost;
			CHECK(r == cc_true && left == 0 && got.at == want_frames * channels && memcmp(out, want, got.at * sizeof(int32_t)) == 0,
			      "callback, %s #%d (%s): %zu of %zu samples after %zu calls", fail_names[kind], nth, crhip_fake_failed(kind) != failed_before ? "injected" : "not reached",
			      got.at, want_frames * channels, calls);
		}

	/* (c) the high-level API: the window keeps what it has pulled; the stream comes out whole however often the device said no */
	for (kind = 0; kind < FAIL_KINDS; ++kind)
		for (nth = 1; nth <= 9; nth += 4)
		{
			static ClownResampler_HighLevel_State hs;
			both io;
			size_t calls = 0;
			cc_bool r = cc_false;
			size_t high_frames = 0;
			int32_t *high_want = (int32_t *)malloc(room * channels * sizeof(int32_t));

			ClownResamplerAMD_Shutdown();
			ClownResamplerAMD_ClearError();
			/* the oracle's own high-level run of the same source */
			{
				oracle_highlevel oh;
				both ref;

				memset(&ref, 0, sizeof(ref));
				ref.src.pcm = pcm + radius * channels;
				ref.src.frames = frames;
				ref.src.channels = channels;
				ref.src.chunk = 0;
				ref.out.out = high_want;
				ref.out.stop_at = (size_t)-1;
				oracle_high_init(&oh, 3, channels, 44100, 48000, 44100);
				if (oracle_high_resample(&oh, table, table_len, pull_both_oracle, sink_both_oracle, &ref))
					oracle_high_end(&oh, table, table_len, sink_both_oracle, &ref);
				high_frames = ref.out.at;
			}
			memset(&io, 0, sizeof(io));
			io.src.pcm = pcm + radius * channels;
			io.src.frames = frames;
			io.src.channels = channels;
			io.src.chunk = 0;
			io.out.out = out;
			io.out.stop_at = (size_t)-1;
			CHECK(ClownResampler_HighLevel_Init(&hs, channels, 44100, 48000, 44100), "HighLevel_Init");
			crhip_fake_fail(kind, nth);
			while (calls < 16)
			{
				r = ClownResampler_HighLevel_Resample(&hs, &pre, pull_both, sink_both, &io);
				disarm();
				++calls;
				if (r)
					break;
				CHECK(ClownResamplerAMD_LastErrorCode() != 0, "high level, %s #%d: cc_false without a stop or an error", fail_names[kind], nth);
				ClownResamplerAMD_ClearError();
			}
			CHECK(r == cc_true, "high level, %s #%d: the source never ran dry", fail_names[kind], nth);
			ClownResampler_HighLevel_ResampleEnd(&hs, &pre, sink_both, &io);
			CHECK(io.out.at == high_frames && memcmp(out, high_want, high_frames * sizeof(int32_t)) == 0, "high level, %s #%d: %zu of %zu samples", fail_names[kind], nth, io.out.at, high_frames);
			ClownResamplerAMD_HighLevel_Release(&hs);
			free(high_want);
		}

	/* (d) variable-rate segments on the "device": 0 frames, the state untouched; the repeat is right */
	{
		const size_t halo = 24;
		ClownResamplerAMD_Segment segments[3] = {{12000, 44100, 48000, 44100}, {8000, 48000, 44100, 44100}, {10000 - 2 * (24 - 3), 44100, 88200, 44100}};
		int mode;

		for (mode = 1; mode <= 2; ++mode)
			for (kind = 0; kind < FAIL_KINDS; ++kind)
				for (nth = 1; nth <= 5; nth += 2)
				{
					ClownResampler_LowLevel_State st, before;
					size_t counts[3], n, total;
					unsigned long long failed_before = crhip_fake_failed(kind);
					int16_t *d_in;
					int32_t *d_out;

					ClownResamplerAMD_Shutdown();
					ClownResamplerAMD_ClearError();
					d_in = (int16_t *)ClownResamplerAMD_DeviceAlloc((frames + 2 * radius) * channels * sizeof(int16_t));
					d_out = (int32_t *)ClownResamplerAMD_DeviceAlloc(room * channels * sizeof(int32_t));
					CHECK(d_in != NULL && d_out != NULL, "DeviceAlloc");
					ClownResamplerAMD_CopyToDevice(d_in, pcm, (frames + 2 * radius) * channels * sizeof(int16_t));
					ClownResamplerAMD_DebugSegmentsMode(mode);
					ClownResampler_LowLevel_Init(&st, channels, 44100, 48000, 44100);
					before = st;
					/* (the timeline's frame 0 lies `halo` frames into the buffer: the first 24 - 3 frames of noise serve as halo) */
					crhip_fake_fail(kind, nth);
					n = ClownResamplerAMD_ResampleSegmentsDevice(&st, &pre, d_in + halo * channels, halo, segments, 3, d_out, room, 0, counts, NULL);
					disarm();
					if (crhip_fake_failed(kind) != failed_before)
					{
						CHECK(n == 0 && memcmp(&st, &before, sizeof(st)) == 0 && ClownResamplerAMD_LastErrorCode() != 0, "segments mode %d, %s #%d: %zu frames, error %d", mode, fail_names[kind], nth, n, ClownResamplerAMD_LastErrorCode());
						ClownResamplerAMD_ClearError();
						n = ClownResamplerAMD_ResampleSegmentsDevice(&st, &pre, d_in + halo * channels, halo, segments, 3, d_out, room, 0, counts, NULL);
					}
					ClownResamplerAMD_StreamSynchronize(NULL);
					total = counts[0] + counts[1] + counts[2];
					CHECK(n != 0 && n == total && ClownResamplerAMD_LastErrorCode() == 0, "segments mode %d after %s #%d: %zu frames (%zu)", mode, fail_names[kind], nth, n, total);
					ClownResamplerAMD_DebugSegmentsMode(0);
					ClownResamplerAMD_DeviceFree(d_in);
					ClownResamplerAMD_DeviceFree(d_out);
				}
	}

	{
		char message[512];
		const int findings = ClownResamplerAMD_DebugSelfCheck(message, sizeof(message));
		CHECK(findings == 0, "after the failures the library is not at rest: %s", message);
	}
	ClownResamplerAMD_SetErrorHandler(on_error, NULL);
	free(pcm);
	free(out);
	free(want);
	printf("ok failures (injected: %llu hipMalloc, %llu hipHostMalloc, %llu launches, %llu copies, %llu synchronises)\n", crhip_fake_failed(0), crhip_fake_failed(1),
	       crhip_fake_failed(2), crhip_fake_failed(3), crhip_fake_failed(4));
	CHECK(crhip_fake_failed(0) >= 8 && crhip_fake_failed(2) >= 8 && crhip_fake_failed(3) >= 4 && crhip_fake_failed(4) >= 8, "the injections did not reach the library");
}

int main(int argc, char **argv)
{
	static const struct
	{
		const char *name;
		void (*run)(void);
	} tests[] = {{"bulk", test_bulk}, {"bulk_batches", test_bulk_batches}, {"callback", test_callback}, {"stream", test_stream}, {"concurrent", test_concurrent},
	             {"segments", test_segments}, {"sharded", test_sharded}, {"device_threads", test_device_threads}, {"failures", test_failures}, {"shutdown", test_shutdown}};
	size_t i, k;

	setenv("CRA_FAKE_DEVICES", "4", 0);
	ClownResamplerAMD_SetErrorHandler(on_error, NULL);
	ClownResampler_Precompute(&pre);
	table_len = oracle_table_len(3);
	table = (int64_t *)malloc(table_len * sizeof(int64_t));
	oracle_precompute(table, 3);
	for (i = 0; i < table_len; ++i)
		if ((int64_t)pre.lanczos_kernel_table[i] != table[i])
		{
			fprintf(stderr, "FAIL: table entry %zu\n", i);
			return 1;
		}
	for (i = 0; i < sizeof(tests) / sizeof(tests[0]); ++i)
	{
		int wanted = argc < 2;

		for (k = 1; k < (size_t)argc; ++k)
			wanted = wanted || strcmp(argv[k], tests[i].name) == 0;
		if (wanted)
			tests[i].run();
	}
	free(table);
	printf(failures == 0 ? "driver: all passed\n" : "driver: %d FAILED\n", failures);
	return failures == 0 ? 0 : 1;
}
