/*
 * clownresampler.h - drop-in interface header of clownresampler_amd.
 *
 * Same types, same ten functions and same call semantics as Clownacy/clownresampler's
 * single header, but this file holds DECLARATIONS ONLY: the implementation is
 * libclownresampler_amd.so, whose per-output-frame windowed-sinc arithmetic
 * (reference clownresampler.h:986-1035 driven by :1058-1092) runs as hand-written
 * HIP kernels on an AMD Instinct MI355X (gfx950).  There is no CPU fallback: if no
 * usable GPU / HIP runtime is present the library reports through the error handler
 * of clownresampler_amd.h, whose default prints the reason and aborts.
 *
 * How a client of the reference switches over (INTEGRATION.md has the details):
 *   - keep its source unchanged; CLOWNRESAMPLER_IMPLEMENTATION / CLOWNRESAMPLER_STATIC
 *     are accepted and ignored (the reference's tests and examples define both,
 *     tests/test-low-level.c:25-28);
 *   - put this directory first on the include path and link -lclownresampler_amd.
 *
 * ABI notes (reference clownresampler.h:480-611, :627-662):
 *   - the integer typedefs below are the reference's DEFAULT (C89) choice; on LP64
 *     that makes cc_s32f/cc_s32l 8 bytes, so the output callback receives 8-byte
 *     samples and ClownResampler_Precomputed is 8 bytes per entry.
 *     libclownresampler_amd.so is built with exactly these.  CC_USE_C99_INTEGERS changes
 *     widths and struct layouts (reference clownresampler.h:483-501); that ABI is the
 *     second build of the same sources, libclownresampler_amd_c99.so - see the #error below.
 *   - CLOWNRESAMPLER_KERNEL_RADIUS changes sizeof(ClownResampler_Precomputed) and the
 *     arithmetic.  The library carries one instance of the API per supported radius;
 *     for a radius other than the default 3 the function names below are redirected to
 *     the instance's symbols (ClownResampler_Precompute -> ClownResampler_Precompute_R8 ...).
 */
#ifndef CLOWNRESAMPLER_AMD_DROPIN_H
#define CLOWNRESAMPLER_AMD_DROPIN_H

#include <stddef.h>

/* ---- configuration (reference clownresampler.h:431-472) ---- */

#ifndef CLOWNRESAMPLER_API
 #define CLOWNRESAMPLER_API /* always external linkage here: the code lives in the shared library */
#endif

#ifndef CLOWNRESAMPLER_KERNEL_RADIUS
 #define CLOWNRESAMPLER_KERNEL_RADIUS 3
#endif
/* Any radius can be built (reference clownresampler.h:445-447: a free compile-time knob) - `make -C clownresampler_amd/csrc
   RADII="3 5 8 <n>"` - and the build writes down which ones it has: a client that asks for another stops here, with the list,
   instead of at link time with an undefined ..._R<n> symbol. */
#if defined(__has_include)
 #if __has_include("clownresampler_amd_radii.h")
  #include "clownresampler_amd_radii.h"
 #endif
#endif

#ifndef CLOWNRESAMPLER_KERNEL_RESOLUTION
 #define CLOWNRESAMPLER_KERNEL_RESOLUTION 0x400
#endif
#if CLOWNRESAMPLER_KERNEL_RESOLUTION != 0x400
 #error "libclownresampler_amd is built for CLOWNRESAMPLER_KERNEL_RESOLUTION 0x400 (the reference default)"
#endif

#ifndef CLOWNRESAMPLER_MAXIMUM_CHANNELS
 #define CLOWNRESAMPLER_MAXIMUM_CHANNELS 16
#endif
#if CLOWNRESAMPLER_MAXIMUM_CHANNELS != 16
 #error "libclownresampler_amd is built for CLOWNRESAMPLER_MAXIMUM_CHANNELS 16 (the reference default)"
#endif

/* ---- integer types (reference clownresampler.h:480-611) ---- */

#ifndef CC_INTEGERS_DEFINED
#define CC_INTEGERS_DEFINED

/* CC_USE_C99_INTEGERS changes the widths of the cc_* types, hence struct layouts and the callback's sample width: it is a
   different ABI.  The library is built in both: libclownresampler_amd.so (C89 types, the reference's default) and
   libclownresampler_amd_c99.so (`make c99` in clownresampler_amd/csrc).  A client that defines CC_USE_C99_INTEGERS must link
   the second one, and says so by defining CLOWNRESAMPLER_AMD_LINKS_C99_LIBRARY as well (CMake target clownresampler_c99). */
#if defined(CC_USE_C99_INTEGERS) && !defined(CLOWNRESAMPLER_AMD_LINKS_C99_LIBRARY) && !defined(CLOWNRESAMPLER_AMD_BUILT_WITH_C99_INTEGERS)
 #error "CC_USE_C99_INTEGERS: link libclownresampler_amd_c99.so (not libclownresampler_amd.so) and define CLOWNRESAMPLER_AMD_LINKS_C99_LIBRARY"
#endif

#if defined(CC_USE_C99_INTEGERS)
/* C99's least / fast types (reference clownresampler.h:483-501).  <inttypes.h> rather than the reference's <stdint.h>: it
   is what defines the PRI* macros the CC_PRI* names below stand for. */
#include <inttypes.h>
typedef int_least8_t cc_s8l;   typedef int_least16_t cc_s16l;   typedef int_least32_t cc_s32l;
typedef uint_least8_t cc_u8l;  typedef uint_least16_t cc_u16l;  typedef uint_least32_t cc_u32l;
typedef int_fast8_t cc_s8f;    typedef int_fast16_t cc_s16f;    typedef int_fast32_t cc_s32f;
typedef uint_fast8_t cc_u8f;   typedef uint_fast16_t cc_u16f;   typedef uint_fast32_t cc_u32f;
/* printf conversions of those types, CC_PRI<conversion><LEAST|FAST><bits> (reference clownresampler.h:503-543), by conversion */
#define CC_PRIdLEAST8 PRIdLEAST8
#define CC_PRIdLEAST16 PRIdLEAST16
#define CC_PRIdLEAST32 PRIdLEAST32
#define CC_PRIdFAST8 PRIdFAST8
#define CC_PRIdFAST16 PRIdFAST16
#define CC_PRIdFAST32 PRIdFAST32
#define CC_PRIiLEAST8 PRIiLEAST8
#define CC_PRIiLEAST16 PRIiLEAST16
#define CC_PRIiLEAST32 PRIiLEAST32
#define CC_PRIiFAST8 PRIiFAST8
#define CC_PRIiFAST16 PRIiFAST16
#define CC_PRIiFAST32 PRIiFAST32
#define CC_PRIuLEAST8 PRIuLEAST8
#define CC_PRIuLEAST16 PRIuLEAST16
#define CC_PRIuLEAST32 PRIuLEAST32
#define CC_PRIuFAST8 PRIuFAST8
#define CC_PRIuFAST16 PRIuFAST16
#define CC_PRIuFAST32 PRIuFAST32
#define CC_PRIoLEAST8 PRIoLEAST8
#define CC_PRIoLEAST16 PRIoLEAST16
#define CC_PRIoLEAST32 PRIoLEAST32
#define CC_PRIoFAST8 PRIoFAST8
#define CC_PRIoFAST16 PRIoFAST16
#define CC_PRIoFAST32 PRIoFAST32
#define CC_PRIxLEAST8 PRIxLEAST8
#define CC_PRIxLEAST16 PRIxLEAST16
#define CC_PRIxLEAST32 PRIxLEAST32
#define CC_PRIxFAST8 PRIxFAST8
#define CC_PRIxFAST16 PRIxFAST16
#define CC_PRIxFAST32 PRIxFAST32
#define CC_PRIXLEAST8 PRIXLEAST8
#define CC_PRIXLEAST16 PRIXLEAST16
#define CC_PRIXLEAST32 PRIXLEAST32
#define CC_PRIXFAST8 PRIXFAST8
#define CC_PRIXFAST16 PRIXFAST16
#define CC_PRIXFAST32 PRIXFAST32
#else
/* C89's types (reference clownresampler.h:545-560): 8- and 16-bit values live in char / short (least) and int (fast), 32-bit
   ones in long either way - 8 bytes on LP64, which is what makes the callback's samples and the table entries 8 bytes wide */
typedef signed char cc_s8l;    typedef signed short cc_s16l;    typedef signed long cc_s32l;
typedef unsigned char cc_u8l;  typedef unsigned short cc_u16l;  typedef unsigned long cc_u32l;
typedef signed int cc_s8f;     typedef signed int cc_s16f;      typedef signed long cc_s32f;
typedef unsigned int cc_u8f;   typedef unsigned int cc_u16f;    typedef unsigned long cc_u32f;
/* whole conversion specifications here, "%" included (reference clownresampler.h:562-602): int-sized below 32 bits, long at 32 */
#define CC_PRIdLEAST8 "%d"
#define CC_PRIdLEAST16 "%d"
#define CC_PRIdFAST8 "%d"
#define CC_PRIdFAST16 "%d"
#define CC_PRIiLEAST8 "%i"
#define CC_PRIiLEAST16 "%i"
#define CC_PRIiFAST8 "%i"
#define CC_PRIiFAST16 "%i"
#define CC_PRIuLEAST8 "%u"
#define CC_PRIuLEAST16 "%u"
#define CC_PRIuFAST8 "%u"
#define CC_PRIuFAST16 "%u"
#define CC_PRIoLEAST8 "%o"
#define CC_PRIoLEAST16 "%o"
#define CC_PRIoFAST8 "%o"
#define CC_PRIoFAST16 "%o"
#define CC_PRIxLEAST8 "%x"
#define CC_PRIxLEAST16 "%x"
#define CC_PRIxFAST8 "%x"
#define CC_PRIxFAST16 "%x"
#define CC_PRIXLEAST8 "%X"
#define CC_PRIXLEAST16 "%X"
#define CC_PRIXFAST8 "%X"
#define CC_PRIXFAST16 "%X"
#define CC_PRIdLEAST32 "%ld"
#define CC_PRIdFAST32 "%ld"
#define CC_PRIiLEAST32 "%li"
#define CC_PRIiFAST32 "%li"
#define CC_PRIuLEAST32 "%lu"
#define CC_PRIuFAST32 "%lu"
#define CC_PRIoLEAST32 "%lo"
#define CC_PRIoFAST32 "%lo"
#define CC_PRIxLEAST32 "%lx"
#define CC_PRIxFAST32 "%lx"
#define CC_PRIXLEAST32 "%lX"
#define CC_PRIXFAST32 "%lX"
#endif

typedef cc_u8l cc_bool;
enum { cc_false = 0, cc_true = 1 };

#endif /* CC_INTEGERS_DEFINED */

/* ---- fixed-point helpers clients may use (reference clownresampler.h:615-625) ---- */

#define CLOWNRESAMPLER_COUNT_OF(x) (sizeof(x) / sizeof(*(x)))
#define CLOWNRESAMPLER_MIN(a, b) ((a) < (b) ? (a) : (b))
#define CLOWNRESAMPLER_MAX(a, b) ((a) > (b) ? (a) : (b))
#define CLOWNRESAMPLER_CLAMP(min, max, x) (CLOWNRESAMPLER_MAX((min), CLOWNRESAMPLER_MIN((max), (x))))
/* 16.16 fixed point (reference clownresampler.h:620-625); note that the conversions to integer and the multiply DIVIDE, so
   for negative values they truncate toward zero - which is the arithmetic the kernels reproduce per tap */
#define CLOWNRESAMPLER_FIXED_POINT_FRACTIONAL_SIZE (1L << 16)
#define CLOWNRESAMPLER_TO_FIXED_POINT_FROM_INTEGER(x) ((x) * CLOWNRESAMPLER_FIXED_POINT_FRACTIONAL_SIZE)
#define CLOWNRESAMPLER_TO_INTEGER_FROM_FIXED_POINT_FLOOR(x) ((x) / CLOWNRESAMPLER_FIXED_POINT_FRACTIONAL_SIZE)
#define CLOWNRESAMPLER_TO_INTEGER_FROM_FIXED_POINT_ROUND(x) (((x) + (CLOWNRESAMPLER_FIXED_POINT_FRACTIONAL_SIZE / 2)) / CLOWNRESAMPLER_FIXED_POINT_FRACTIONAL_SIZE)
#define CLOWNRESAMPLER_TO_INTEGER_FROM_FIXED_POINT_CEILING(x) (((x) + (CLOWNRESAMPLER_FIXED_POINT_FRACTIONAL_SIZE - 1)) / CLOWNRESAMPLER_FIXED_POINT_FRACTIONAL_SIZE)
#define CLOWNRESAMPLER_FIXED_POINT_MULTIPLY(a, b) ((a) * (b) / CLOWNRESAMPLER_FIXED_POINT_FRACTIONAL_SIZE)

/* ---- caller-owned state (reference clownresampler.h:627-662; layouts must match field for field,
        callers read e.g. lowest_level.integer_stretched_kernel_radius to size their padding) ---- */

typedef struct ClownResampler_Precomputed
{
	cc_s32l lanczos_kernel_table[CLOWNRESAMPLER_KERNEL_RADIUS * 2 * CLOWNRESAMPLER_KERNEL_RESOLUTION];
} ClownResampler_Precomputed;

typedef struct ClownResampler_LowestLevel_Configuration
{
	size_t stretched_kernel_radius;         /* 16.16 */
	size_t integer_stretched_kernel_radius; /* frames of padding needed each side of the input */
	size_t stretched_kernel_radius_delta;   /* 16.16 */
	size_t kernel_step_size;
} ClownResampler_LowestLevel_Configuration;

typedef struct ClownResampler_LowLevel_State
{
	ClownResampler_LowestLevel_Configuration lowest_level;
	cc_u8f channels;
	size_t position_integer;
	cc_u32f position_fractional;            /* 16.16 */
	cc_u32f increment;                      /* 16.16 */
} ClownResampler_LowLevel_State;

typedef struct ClownResampler_HighLevel_State
{
	ClownResampler_LowLevel_State low_level;
	cc_s16l input_buffer[0x1000];
	cc_s16l *input_buffer_start;
	cc_s16l *input_buffer_end;
	size_t maximum_integer_stretched_kernel_radius;
	size_t leading_padding_frames_needed, trailing_padding_frames_remaining;
} ClownResampler_HighLevel_State;

typedef size_t (*ClownResampler_InputCallback)(void *user_data, cc_s16l *buffer, size_t total_frames);
typedef cc_bool (*ClownResampler_OutputCallback)(void *user_data, const cc_s32f *frame, cc_u8f total_samples);

/* ---- per-radius symbol redirection ---- */

#define CLOWNRESAMPLER_AMD_CAT2(a, b) a##b
#define CLOWNRESAMPLER_AMD_CAT(a, b) CLOWNRESAMPLER_AMD_CAT2(a, b)
#define CLOWNRESAMPLER_AMD_SYM(name) CLOWNRESAMPLER_AMD_CAT(CLOWNRESAMPLER_AMD_CAT(name, _R), CLOWNRESAMPLER_KERNEL_RADIUS)

#if CLOWNRESAMPLER_KERNEL_RADIUS != 3
 #define ClownResampler_Precompute            CLOWNRESAMPLER_AMD_SYM(ClownResampler_Precompute)
 #define ClownResampler_LowestLevel_Configure CLOWNRESAMPLER_AMD_SYM(ClownResampler_LowestLevel_Configure)
 #define ClownResampler_LowestLevel_Resample  CLOWNRESAMPLER_AMD_SYM(ClownResampler_LowestLevel_Resample)
 #define ClownResampler_LowLevel_Init         CLOWNRESAMPLER_AMD_SYM(ClownResampler_LowLevel_Init)
 #define ClownResampler_LowLevel_Adjust       CLOWNRESAMPLER_AMD_SYM(ClownResampler_LowLevel_Adjust)
 #define ClownResampler_LowLevel_Resample     CLOWNRESAMPLER_AMD_SYM(ClownResampler_LowLevel_Resample)
 #define ClownResampler_HighLevel_Init        CLOWNRESAMPLER_AMD_SYM(ClownResampler_HighLevel_Init)
 #define ClownResampler_HighLevel_Resample    CLOWNRESAMPLER_AMD_SYM(ClownResampler_HighLevel_Resample)
 #define ClownResampler_HighLevel_Adjust      CLOWNRESAMPLER_AMD_SYM(ClownResampler_HighLevel_Adjust)
 #define ClownResampler_HighLevel_ResampleEnd CLOWNRESAMPLER_AMD_SYM(ClownResampler_HighLevel_ResampleEnd)
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- common (reference clownresampler.h:682) ---- */

/* Fills the Lanczos table; same values as the reference's (libm sin, truncation toward zero). Host-side. */
CLOWNRESAMPLER_API void ClownResampler_Precompute(ClownResampler_Precomputed *precomputed);

/* ---- lowest level (reference clownresampler.h:687-688) ---- */

CLOWNRESAMPLER_API cc_bool ClownResampler_LowestLevel_Configure(ClownResampler_LowestLevel_Configuration *configuration, cc_u32f input_sample_rate, cc_u32f output_sample_rate, cc_u32f low_pass_filter_sample_rate);

/* One output frame, ACCUMULATED INTO output_frame and then normalised as a whole, as in the reference.
   Runs one (tiny) GPU launch: meant for compatibility, not speed. */
CLOWNRESAMPLER_API void ClownResampler_LowestLevel_Resample(const ClownResampler_LowestLevel_Configuration *configuration, const ClownResampler_Precomputed *precomputed, cc_s32f *output_frame, cc_u8f channels, const cc_s16l *input_buffer, size_t position_integer, cc_u32f position_fractional);

/* ---- low level (reference clownresampler.h:711, :719, :749) ---- */

#ifndef CLOWNRESAMPLER_NO_LOW_LEVEL_API
CLOWNRESAMPLER_API cc_bool ClownResampler_LowLevel_Init(ClownResampler_LowLevel_State *resampler, cc_u8f channels, cc_u32f input_sample_rate, cc_u32f output_sample_rate, cc_u32f low_pass_filter_sample_rate);
CLOWNRESAMPLER_API cc_bool ClownResampler_LowLevel_Adjust(ClownResampler_LowLevel_State *resampler, cc_u32f input_sample_rate, cc_u32f output_sample_rate, cc_u32f low_pass_filter_sample_rate);

/* input_buffer points at the START of the left padding (integer_stretched_kernel_radius frames each side, not
   counted in *total_input_frames).  Returns cc_true when the input ran out, cc_false when output_callback
   returned 0; *total_input_frames and the state are left exactly as the reference leaves them, so a caller can
   resume.  The frames are computed on the GPU in growing batches and replayed through output_callback on the
   calling thread, in order. */
CLOWNRESAMPLER_API cc_bool ClownResampler_LowLevel_Resample(ClownResampler_LowLevel_State *resampler, const ClownResampler_Precomputed *precomputed, const cc_s16l *input_buffer, size_t *total_input_frames, ClownResampler_OutputCallback output_callback, const void *user_data);
#endif

/* ---- high level (reference clownresampler.h:770, :825, :839, :847) ---- */

#ifndef CLOWNRESAMPLER_NO_HIGH_LEVEL_API
CLOWNRESAMPLER_API cc_bool ClownResampler_HighLevel_Init(ClownResampler_HighLevel_State *resampler, cc_u8f channels, cc_u32f input_sample_rate, cc_u32f output_sample_rate, cc_u32f low_pass_filter_sample_rate);
CLOWNRESAMPLER_API cc_bool ClownResampler_HighLevel_Resample(ClownResampler_HighLevel_State *resampler, const ClownResampler_Precomputed *precomputed, ClownResampler_InputCallback input_callback, ClownResampler_OutputCallback output_callback, const void *user_data);
#ifndef CLOWNRESAMPLER_NO_HIGH_LEVEL_ADJUST
CLOWNRESAMPLER_API cc_bool ClownResampler_HighLevel_Adjust(ClownResampler_HighLevel_State *resampler, cc_u32f input_sample_rate, cc_u32f output_sample_rate, cc_u32f low_pass_filter_sample_rate);
#endif
#ifndef CLOWNRESAMPLER_NO_HIGH_LEVEL_RESAMPLE_END
CLOWNRESAMPLER_API cc_bool ClownResampler_HighLevel_ResampleEnd(ClownResampler_HighLevel_State *resampler, const ClownResampler_Precomputed *precomputed, ClownResampler_OutputCallback output_callback, const void *user_data);
#endif
#endif

#ifdef __cplusplus
}
#endif

#endif /* CLOWNRESAMPLER_AMD_DROPIN_H */
