/*
 * cr_oracle.h - CPU restatement of clownresampler's windowed-sinc path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (clownresampler_amd/,
 * include/) may include, link or call this.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, as the checker.
 *
 * Parity status: PINNED.  oracle/ref/ builds the real reference header in
 * place (oracle/_ref/libclownref_r{3,8}.so) and tests/test_oracle_vs_ref.py
 * compares this restatement with it bit for bit (table, ratio/config scalars,
 * full streams, chunked + early-stop resume, high-level API); the outputs of
 * that real reference are committed as fixtures under tests/golden/ and
 * tests/test_oracle_golden.py checks the restatement against them on the GPU
 * box, where /root/reference does not exist.
 *
 * Integer widths follow the reference's default (C89) typedefs on LP64, the
 * platform the reference was built on to make the fixtures:
 *   cc_s32l / cc_s32f = long  -> int64_t      (clownresampler.h:548,556)
 *   cc_u32f = unsigned long   -> uint64_t     (clownresampler.h:560)
 *   cc_u8f  = unsigned int    -> uint32_t     (clownresampler.h:558)
 *   cc_s16l = short           -> int16_t      (clownresampler.h:547)
 *   cc_bool = unsigned char   -> uint8_t      (clownresampler.h:606)
 * Struct layouts are byte-identical to the reference's on LP64
 * (clownresampler.h:632-659) so one ctypes definition serves the oracle, the
 * compiled reference and the product library.
 *
 * The kernel radius (CLOWNRESAMPLER_KERNEL_RADIUS, clownresampler.h:445-447,
 * compile-time in the reference) is a run-time argument here.
 */
#ifndef CR_ORACLE_H
#define CR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_FRAC_ONE 65536           /* 16.16, clownresampler.h:620 */
#define ORACLE_TABLE_RES 1024           /* CLOWNRESAMPLER_KERNEL_RESOLUTION, clownresampler.h:452-454 */
#define ORACLE_MAX_CHANNELS 16          /* CLOWNRESAMPLER_MAXIMUM_CHANNELS, clownresampler.h:458-460 */
#define ORACLE_STAGING_SAMPLES 0x1000   /* clownresampler.h:654 */

/* clownresampler.h:632-638 */
typedef struct oracle_config
{
	uint64_t stretched_radius;      /* 16.16 */
	uint64_t radius_frames;         /* integer_stretched_kernel_radius */
	uint64_t radius_delta;          /* 16.16 */
	uint64_t table_step;            /* kernel_step_size */
} oracle_config;

/* clownresampler.h:640-648 */
typedef struct oracle_lowlevel
{
	oracle_config cfg;
	uint32_t channels;
	uint64_t pos_int;
	uint64_t pos_frac;              /* 16.16 */
	uint64_t increment;             /* 16.16 */
} oracle_lowlevel;

/* clownresampler.h:650-659 */
typedef struct oracle_highlevel
{
	oracle_lowlevel low;
	int16_t staging[ORACLE_STAGING_SAMPLES];
	int16_t *win_begin;
	int16_t *win_end;
	uint64_t max_radius_frames;
	uint64_t lead_needed;
	uint64_t trail_left;
} oracle_highlevel;

/* clownresampler.h:661-662 */
typedef size_t (*oracle_input_cb)(void *user, int16_t *buffer, size_t max_frames);
typedef uint8_t (*oracle_output_cb)(void *user, const int64_t *frame, uint32_t samples);

/* Normalisation variants.  CURRENT is what the shipped header does
 * (clownresampler.h:1025-1033).  LEGACY_GAIN replaces only that last step by a
 * constant 16.16 gain ratio(lowpass, in): it is what generated the reference's
 * stale goldens tests/test3 / tests/test4 (finding 1) and
 * exists so those files can pin everything before the normalisation. */
enum { ORACLE_NORM_CURRENT = 0, ORACLE_NORM_LEGACY_GAIN = 1 };

size_t   oracle_table_len(unsigned radius);
void     oracle_precompute(int64_t *table, unsigned radius);
uint64_t oracle_ratio(uint64_t a, uint64_t b);
uint8_t  oracle_configure(oracle_config *cfg, unsigned radius, uint64_t in_rate, uint64_t out_rate, uint64_t lowpass_rate);
void     oracle_frame(const oracle_config *cfg, const int64_t *table, size_t table_len, int64_t *accum,
                      uint32_t channels, const int16_t *padded_in, uint64_t pos_int, uint64_t pos_frac);

uint8_t  oracle_low_init(oracle_lowlevel *st, unsigned radius, uint32_t channels, uint64_t in_rate, uint64_t out_rate, uint64_t lowpass_rate);
uint8_t  oracle_low_adjust(oracle_lowlevel *st, unsigned radius, uint64_t in_rate, uint64_t out_rate, uint64_t lowpass_rate);
uint8_t  oracle_low_resample(oracle_lowlevel *st, const int64_t *table, size_t table_len, const int16_t *padded_in,
                             size_t *frames_left, oracle_output_cb emit, const void *user);

uint8_t  oracle_high_init(oracle_highlevel *st, unsigned radius, uint32_t channels, uint64_t in_rate, uint64_t out_rate, uint64_t lowpass_rate);
uint8_t  oracle_high_resample(oracle_highlevel *st, const int64_t *table, size_t table_len, oracle_input_cb pull,
                              oracle_output_cb emit, const void *user);
uint8_t  oracle_high_adjust(oracle_highlevel *st, unsigned radius, uint64_t in_rate, uint64_t out_rate, uint64_t lowpass_rate);
uint8_t  oracle_high_end(oracle_highlevel *st, const int64_t *table, size_t table_len, oracle_output_cb emit, const void *user);

/* ---- harness conveniences (not in the reference; thin loops over the above) ---- */

/* Closed-form number of frames a fresh-or-carried state emits for `frames` input frames. */
uint64_t oracle_count_output_frames(const oracle_lowlevel *st, uint64_t frames);

/* Runs oracle_low_resample with a callback that stores each sample as int32
 * (the reference harness's on-disk format, tests/test-low-level.c:43-49) into
 * `out` (capacity in frames).  Stops early, exactly like a callback returning
 * 0, once capacity is reached.  Returns frames written.  norm_mode selects the
 * normalisation variant above. */
size_t oracle_low_resample_i32(oracle_lowlevel *st, const int64_t *table, size_t table_len, const int16_t *padded_in,
                               size_t *frames_left, int32_t *out, size_t out_capacity_frames, int norm_mode,
                               uint64_t legacy_gain, uint8_t *ran_out_of_input);

/* High-level one-shot over an in-memory PCM buffer: Init state is supplied by
 * the caller; pulls `pull_chunk` frames at most per input callback (0 = as many
 * as asked), runs oracle_high_resample then oracle_high_end.  Returns frames written. */
size_t oracle_high_run_i32(oracle_highlevel *st, const int64_t *table, size_t table_len, const int16_t *pcm,
                           size_t pcm_frames, size_t pull_chunk, int32_t *out, size_t out_capacity_frames);

/* Stream hash of SURVEY.md section 8(d): h = (h ^ (uint64)(uint32)sample) * 1099511628211. */
uint64_t oracle_stream_hash(const int32_t *samples, size_t count, uint64_t seed);

/* Synthetic PCM of SURVEY.md section 8(d): xorshift64 white noise, sample = (int16)(x >> 48). */
uint64_t oracle_fill_noise(int16_t *dst, size_t samples, uint64_t state);

/* Multi-threaded CPU baseline: splits the input-frame
 * range over `threads` independent states (closed-form start state), each
 * running oracle_low_resample through a storing callback.  out must hold
 * oracle_count_output_frames() frames.  Returns frames written. */
size_t oracle_low_resample_i32_mt(const oracle_lowlevel *fresh, const int64_t *table, size_t table_len,
                                  const int16_t *padded_in, size_t frames, int32_t *out, unsigned threads);

#ifdef __cplusplus
}
#endif

#endif /* CR_ORACLE_H */
