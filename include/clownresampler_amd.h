/*
 * clownresampler_amd.h - extension entry points of libclownresampler_amd.so (NOT in the reference).
 *
 * The reference's hot path delivers one frame per indirect call (clownresampler.h:662, :1081); that
 * shape is kept for compatibility in include/clownresampler.h, but it can never run at memory speed.
 * These are the callback-free forms of the SAME function: they produce exactly the frames
 * ClownResampler_LowLevel_Resample (reference clownresampler.h:1058-1092) would hand to a callback that
 * stores them and returns 0 once `output_capacity_frames` frames were stored, in the on-disk layout of the
 * reference's harness (int32, frame-major, channel-minor: tests/test-low-level.c:43-49), and leave
 * *total_input_frames and the state exactly as that call would.
 *
 * Everything here is a plain C ABI: pointers, sizes and the reference's own POD structs.  Device pointers
 * and the HIP stream are passed as void*.
 *
 * Include after (or instead of) clownresampler.h; CLOWNRESAMPLER_KERNEL_RADIUS selects the instance as
 * it does there.
 */
#ifndef CLOWNRESAMPLER_AMD_EXT_H
#define CLOWNRESAMPLER_AMD_EXT_H

#include <stdint.h>

#include "clownresampler.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------
 * Errors.  The reference's resample calls have no error channel (their cc_bool means "ran out of
 * input" vs "callback stopped", clownresampler.h:746-748), and this library has no CPU fallback, so a
 * runtime failure (no GPU, HIP error, out of device memory, unsupported argument) goes to a handler.
 * With NO handler installed the failure is recorded for the calling thread (ClownResamplerAMD_LastErrorCode / LastErrorMessage), the
 * message is written to stderr (the first eight of a process), and the failed call returns - it never ends the process: the entry
 * point reports the reference's "callback stopped" outcome (cc_false; ClownResampler_LowLevel_ResampleBulk and the device entry
 * points: 0 frames, *ran_out_of_input = cc_false) with the state and *total_input_frames as they are after the frames the consumer
 * HAS been given (none: untouched), so the client still owns its input and may try again or fall back.  The same holds after a
 * handler that returns.  CLOWNRESAMPLER_AMD_ABORT_ON_ERROR=1 in the environment (read at the first failure) makes the handler-less
 * default print, dump the flight recorder and abort() instead.
 * ------------------------------------------------------------------------------------------- */
enum
{
	CLOWNRESAMPLER_AMD_OK = 0,
	CLOWNRESAMPLER_AMD_ERROR_NO_DEVICE = 1,     /* no HIP device / runtime unusable */
	CLOWNRESAMPLER_AMD_ERROR_HIP = 2,           /* a HIP call failed; message carries hipGetErrorString */
	CLOWNRESAMPLER_AMD_ERROR_ARGUMENT = 3,      /* e.g. channels == 0, weight sum of 0 (the reference divides by zero there) */
	CLOWNRESAMPLER_AMD_ERROR_PLAN_MISMATCH = 4  /* state does not match the plan it is used with */
};

typedef void (*ClownResamplerAMD_ErrorHandler)(int code, const char *message, void *user_data);

void ClownResamplerAMD_SetErrorHandler(ClownResamplerAMD_ErrorHandler handler, void *user_data); /* NULL restores the default */
int ClownResamplerAMD_LastErrorCode(void);              /* of the calling thread; reset by ClownResamplerAMD_ClearError */
const char *ClownResamplerAMD_LastErrorMessage(void);
void ClownResamplerAMD_ClearError(void);

/* ---------------------------------------------------------------------------------------------
 * Device selection and lifetime.  The reference API has no Deinit, so GPU resources live in a
 * process-wide cache created on first use (per device: staging buffers, streams, uploaded tables and
 * plans keyed by table contents + configuration) and released by Shutdown or at exit.
 * ------------------------------------------------------------------------------------------- */
/* sha256 (first 16 hex digits) over the sources this library was built from: ties profiles to builds */
const char *ClownResamplerAMD_BuildId(void);
/* the values of CLOWNRESAMPLER_KERNEL_RADIUS this library was built for, space-separated ("3 5 8"; csrc/Makefile RADII) */
const char *ClownResamplerAMD_BuiltRadii(void);
int ClownResamplerAMD_DeviceCount(void);                /* 0 when there is none; never calls the error handler */
/* 1 when the resampling entry points of this library can run in this process - a gfx950 (MI355X) device is visible to the calling thread's
   selection - else 0.  Never calls the error handler, never aborts: the question to ask ONCE, at start-up, by a client that wants to fall
   back on the reference's own header (clownresampler.h with CLOWNRESAMPLER_IMPLEMENTATION: same API, same results, one core) where
   there is no such device, instead of finding out from a failed first ClownResampler_*_Resample call.  This
   library itself has no CPU path by design. */
int ClownResamplerAMD_IsUsable(void);
int ClownResamplerAMD_SetDevice(int ordinal);           /* process default: device used by calls of threads that have not chosen one; 0 on success */
int ClownResamplerAMD_SetThreadDevice(int ordinal);     /* device used by subsequent calls of the CALLING THREAD (-1: follow the process default again) */
int ClownResamplerAMD_GetDevice(void);                  /* what the calling thread's next call would use */
/* Every device keeps its own context (plans, ticket blocks, staging buffers and streams) from first use on: changing the
   current device tears nothing down, calls on different devices do not serialise each other, and a plan remembers its
   device (ClownResamplerAMD_ResampleDevice launches there whatever the current device is).  Shutdown releases everything
   on every device; it needs a quiescent library (no resample call in progress on any thread). */
void ClownResamplerAMD_Shutdown(void);
/* Launches issued while their stream is being captured into a hipGraph draw their scheduling scratch ("ticket blocks") from
   a pool that is never recycled, because the graph may be replayed at any later time; device memory cannot be allocated
   during a capture, so the pool is sized ahead: room for 256 captured launches per device by default, more with this call
   (before the capture; current device).  0 on success. */
int ClownResamplerAMD_ReserveCaptureLaunches(size_t launches);
/* The other end of that pool, for clients that re-capture graphs over time: call it once every graph that captured launches of
   this library on the current device has been destroyed (none is executing or will be launched again).  Every block handed
   out to a captured launch becomes available again and the memory of earlier ReserveCaptureLaunches calls is returned.
   0 on success. */
int ClownResamplerAMD_ReleaseCapturedLaunches(void);

/* High-level (streaming) API: how many input frames ClownResampler_HighLevel_Resample may collect from the input callback
   before it resamples them in one GPU call (default 262144).  The reference's 0x1000-sample staging buffer
   (clownresampler.h:654) corresponds to 0: one input pull per GPU call - exact call-for-call behaviour, but every call
   then pays a launch and two PCIe copies for ~2000 frames.  The output is identical for every setting; what changes is
   how far ahead of the output the input callback is asked for data (the window grows 4x per refill only while the
   consumer keeps draining whole windows, and falls back to one pull as soon as the output callback stops a call). */
void ClownResamplerAMD_SetStreamingWindow(size_t frames);
/* The side window of a high-level state lives in the library, registered under the state's address (the reference API has no
   Deinit).  A state initialised at an address that had a window before takes that window over, so programs that create and
   discard resamplers do not accumulate windows; this call frees the window of a state that will not be used again (the state
   needs a new ClownResampler_HighLevel_Init afterwards).  ClownResamplerAMD_Shutdown frees them all. */
void ClownResamplerAMD_HighLevel_Release(ClownResampler_HighLevel_State *resampler);
size_t ClownResamplerAMD_StreamingWindowCount(void);    /* side windows the library holds at the moment */

/* Thin helpers so a C client needs no HIP headers (current device; ...On: a given device). */
void *ClownResamplerAMD_DeviceAllocOn(int device, size_t bytes);
void *ClownResamplerAMD_DeviceAlloc(size_t bytes);
void ClownResamplerAMD_DeviceFree(void *device_pointer);
int ClownResamplerAMD_CopyToDevice(void *device_destination, const void *host_source, size_t bytes);
int ClownResamplerAMD_CopyFromDevice(void *host_destination, const void *device_source, size_t bytes);
int ClownResamplerAMD_StreamSynchronize(void *hip_stream);

/* ---------------------------------------------------------------------------------------------
 * Closed forms of the output-timeline walk (reference clownresampler.h:1058-1092).  Host-only, no GPU.
 * ------------------------------------------------------------------------------------------- */

/* Frames ClownResampler_LowLevel_Resample emits for `total_input_frames` from `state` if never stopped. */
size_t ClownResamplerAMD_CountOutputFrames(const ClownResampler_LowLevel_State *state, size_t total_input_frames);

/* Advances the position of `state` as if `frames` output frames had been emitted (clownresampler.h:1076-1078, n times). */
void ClownResamplerAMD_AdvanceState(ClownResampler_LowLevel_State *state, size_t frames);

/* One shard of an output timeline split into `shard_count` contiguous blocks of output frames (the last may be
   shorter).  Filled from a state and the total number of input frames of the whole stream. */
typedef struct ClownResamplerAMD_Shard
{
	size_t first_output_frame;      /* index in the whole stream's output */
	size_t output_frames;           /* frames this shard produces */
	size_t first_input_frame;       /* offset, in frames, to add to the whole stream's padded input pointer */
	size_t input_frames;            /* value of *total_input_frames for the shard's call (excludes the halo); the call must
	                                   also be given output_capacity_frames == output_frames, which is what ends it */
	size_t halo_frames;             /* integer_stretched_kernel_radius: real neighbour frames needed each side */
	ClownResampler_LowLevel_State state; /* state to run the shard with */
} ClownResamplerAMD_Shard;

int ClownResamplerAMD_PlanShard(const ClownResampler_LowLevel_State *state, size_t total_input_frames, unsigned shard, unsigned shard_count, ClownResamplerAMD_Shard *out);

/* ---------------------------------------------------------------------------------------------
 * Bulk resampling, host buffers (radius-specific: redirected like the reference functions).
 * ------------------------------------------------------------------------------------------- */
#if CLOWNRESAMPLER_KERNEL_RADIUS != 3
 #define ClownResamplerAMD_ResampleShardedDevice CLOWNRESAMPLER_AMD_SYM(ClownResamplerAMD_ResampleShardedDevice)
 #define ClownResampler_LowLevel_ResampleBulk CLOWNRESAMPLER_AMD_SYM(ClownResampler_LowLevel_ResampleBulk)
 #define ClownResampler_LowLevel_ResampleBulkS16 CLOWNRESAMPLER_AMD_SYM(ClownResampler_LowLevel_ResampleBulkS16)
 #define ClownResamplerAMD_PlanCreate          CLOWNRESAMPLER_AMD_SYM(ClownResamplerAMD_PlanCreate)
 #define ClownResamplerAMD_ResampleSegmentsDevice CLOWNRESAMPLER_AMD_SYM(ClownResamplerAMD_ResampleSegmentsDevice)
#endif

/* Same arguments as ClownResampler_LowLevel_Resample with (output, output_capacity_frames) in place of the
   callback.  Returns the number of frames written; *ran_out_of_input (may be NULL) receives what the reference
   call would have returned.  Synchronous: input is uploaded, output downloaded. */
size_t ClownResampler_LowLevel_ResampleBulk(ClownResampler_LowLevel_State *resampler, const ClownResampler_Precomputed *precomputed, const cc_s16l *input_buffer, size_t *total_input_frames, int32_t *output, size_t output_capacity_frames, cc_bool *ran_out_of_input);

/* The same with the 16-bit clamp every consumer of the reference applies in its output callback fused in: each sample is
   clamped to [-0x7FFF, 0x7FFF] (sic: examples/low-level.c:69-80, examples/high-level.c:74-85) and stored as int16.
   Opt-in: the reference's own output is the unclamped int32 of ClownResampler_LowLevel_ResampleBulk. */
size_t ClownResampler_LowLevel_ResampleBulkS16(ClownResampler_LowLevel_State *resampler, const ClownResampler_Precomputed *precomputed, const cc_s16l *input_buffer, size_t *total_input_frames, int16_t *output, size_t output_capacity_frames, cc_bool *ran_out_of_input);

/* ---------------------------------------------------------------------------------------------
 * Bulk resampling, device-resident buffers.
 * ------------------------------------------------------------------------------------------- */
typedef struct ClownResamplerAMD_Plan ClownResamplerAMD_Plan;

typedef struct ClownResamplerAMD_PlanInfo
{
	uint32_t kernel;            /* 6 = k_wave2s (k_wave2's arithmetic with one lane per channel PAIR of a frame: wide frames, long windows), 1 = k_poly (polyphase rows in LDS, workgroup tiles), 2 = k_wave (same, wave-autonomous), 3 = k_up / k_up2 (input-stationary, strong upsampling), 4 = k_wave2 (wave-autonomous, expanded window, 64-bit multiply-add taps), 0 = generic 64-bit kernel */
	uint32_t channels;
	uint32_t slots;             /* taps evaluated per output frame (zero-weight slots included) */
	uint32_t first_slot;        /* frame offset of slot 0 relative to position_integer, in padded-buffer frames (row_mode 0: of the
	                               phases with the smallest min_relative; the rows of the others start that much later) */
	uint32_t rows;              /* polyphase rows */
	uint32_t row_stride;        /* int32 per row: slots weights, then the 17.15 reciprocal, then padding */
	uint32_t row_mode;          /* 0 = affine row index, 1 = pure-upsampling row index ((65536 - frac) >> 6) */
	uint32_t threads;           /* workgroup size */
	uint32_t tile_frames;       /* output frames per LDS tile */
	uint32_t lds_bytes;         /* dynamic LDS per workgroup */
	uint32_t max_blocks;        /* persistent grid size used for large launches */
	uint32_t specialised;       /* 1 when a (channels, slots) template instance is used */
	uint32_t variant;           /* tuning variant the plan was built for (0xFFFF: the instance's measured default) */
	uint32_t norm_mode;         /* 0: |accumulator x reciprocal| < 2^31 for every row (signed multiply), 1: < 2^32 (on magnitudes) */
	uint32_t brief_kernel;      /* kernel 3 only: the kernel (numbered as `kernel`) that launches of fewer than brief_below output frames take instead - k_up's wave-tiles are too long for them - or 0 */
	uint32_t brief_variant;     /* ... and its tuning variant */
	uint64_t brief_below;       /* 0: every launch takes `kernel` */
} ClownResamplerAMD_PlanInfo;

/* Builds (or fetches from the cache) the device-side plan for the configuration, channel count and increment
   of `state` and the contents of `precomputed`: the polyphase weight rows with their reciprocals, uploaded to
   the current device.  The plan stays valid until ClownResamplerAMD_Shutdown. */
ClownResamplerAMD_Plan *ClownResamplerAMD_PlanCreate(const ClownResampler_LowLevel_State *state, const ClownResampler_Precomputed *precomputed);
void ClownResamplerAMD_PlanGetInfo(const ClownResamplerAMD_Plan *plan, ClownResamplerAMD_PlanInfo *info);

/* Plans are cached by the CONTENTS of (table, configuration, channels, increment); plans that differ only in their
   increment share one copy of the rows.  Plans made by the library on behalf of the reference-signature calls are
   dropped, least recently used first, once there are more than `plans` of them (default 64; their device memory is
   released with hipFree, which waits for the device).  Plans returned by ClownResamplerAMD_PlanCreate are never dropped. */
void ClownResamplerAMD_SetPlanCacheLimit(size_t plans);
size_t ClownResamplerAMD_PlanCacheCount(void);

/* The kernel a launch from this fractional position takes, numbered as ClownResamplerAMD_PlanInfo.kernel plus 5 = k_int: at a
   whole-number downsampling ratio (2:1, 3:1, 4:1, 6:1; 1 to 8 channels, 2:1 up to 16) every frame of a launch uses ONE polyphase row, which
   then travels in the kernel arguments - if that row has the slot signs the instance was built for (always, for a stream that
   starts from ClownResampler_LowLevel_Init; a stream resumed at another fraction may not).  The same for the ratios whose
   fractional position repeats after 2 or 4 frames (3:2; with the 5- and 8-lobe builds also 1:2 and 1:4): the rows of the period
   travel, for a launch that starts at the phase the instance begins with - a long launch that starts elsewhere in the period
   gives its first frames to the ordinary kernel, which is what this function then names.  Ignores launch-length rules
   (brief_below). */
uint32_t ClownResamplerAMD_PlanKernelAt(const ClownResamplerAMD_Plan *plan, uint32_t position_fractional);
/* Launches enqueued by this process so far on `kernel` (0 ... 6, numbered as above; 8 = k_seg): lets tests and benchmarks assert that the
   kernel they mean is the one that ran.  7 = how many of those launches drew their tiles as tickets (k_poly's and k_int's long
   launches; a launch of fewer than eight tiles per workgroup is dealt round-robin instead). */
unsigned long long ClownResamplerAMD_DebugLaunchCount(unsigned kernel);
/* Test hook: whole-number ratios take the plan's ordinary kernel instead of k_int (the A/B leg; also CLOWNRESAMPLER_AMD_NO_INT_KERNEL
   in the environment at first use, which additionally skips k_int's one-time setup). */
void ClownResamplerAMD_DebugDisableIntKernel(int on);
/* 0, or the kernel (numbered as ClownResamplerAMD_PlanInfo.kernel: 1 = k_poly, 4 = k_wave2) of the STEREO instance that long launches
   of this MONO plan run on as "dual mono": output frames j and j + H, whose fractional positions are equal, as its two channels. */
uint32_t ClownResamplerAMD_PlanDualMonoKernel(const ClownResamplerAMD_Plan *plan);
/* 8 (the number ClownResamplerAMD_DebugLaunchCount counts k_seg under) when LONG launches of this plan may take k_seg - stereo, 15 slots with
   the instance's slot signs, 4x to 16x upsampling; which launch does is the launch-length rule's to say (ClownResamplerAMD_DebugSegKernel) -
   else 0.  ClownResamplerAMD_PlanInfo.kernel names the kernel of the plan's ordinary launches. */
uint32_t ClownResamplerAMD_PlanSegKernel(const ClownResamplerAMD_Plan *plan);
/* 1 when the plan's k_poly instance computes from PADDED tiles: 9-11 and 13-15 channels without a specialised instance, up to 2:1
   downsampling - a frame of 18 to 30 bytes leaves a lane's share of it on any 2-byte boundary, so every tile is repacked once,
   LDS -> LDS, to frames of 32 bytes whose shares are one aligned 16-byte read per tap (CLOWNRESAMPLER_AMD_NO_PADDED_TILES in the
   environment at first use: never). */
uint32_t ClownResamplerAMD_PlanPaddedTiles(const ClownResamplerAMD_Plan *plan);
/* Test hook for k_seg (long stereo launches of 4x - 16x upsampling with 8 lobes - increments 4096 ... 16384, CR_SEG_MIN_INCREMENT and the instance's ring: the lanes of a wave on output frames of equal fraction,
   the polyphase row in scalar registers): 0 = the rule (launches whose last, partial block of 64 segments wastes little), 1 = every
   launch the kernel can take, 2 = never (also CLOWNRESAMPLER_AMD_NO_SEG in the environment at first use).  Kernel 8 of
   ClownResamplerAMD_DebugLaunchCount counts its launches. */
void ClownResamplerAMD_DebugSegKernel(int mode);
/* Test hook (host only, no device needed): the predicate that keeps dual-mono launches inside their kernels' 32-bit buffer-descriptor arithmetic. */
int ClownResamplerAMD_DebugDualMonoFits(uint64_t n_out, uint64_t half, uint64_t tile_frames, uint64_t increment, uint64_t in_valid_bytes);
/* Test hook: long MONO launches stay on the plan's mono kernel instead of running as two phase-aligned "channels" of the stereo
   instance (dual mono; also CLOWNRESAMPLER_AMD_NO_DUAL_MONO in the environment at first use). */
void ClownResamplerAMD_DebugDisableDualMono(int on);
/* Test hook for ClownResamplerAMD_ResampleSegmentsDevice: 0 = the measured rule picks (default), 1 = always one launch per segment
   (the polyphase kernels), 2 = always ONE launch for all segments (the generic kernel with a segment table). */
void ClownResamplerAMD_DebugSegmentsMode(int mode);

/* 1 when the host-pointer entry points would work on [host, host + bytes) in place - page-locked memory the current device can address
   (hipHostMalloc, hipHostRegister) - instead of staging it (pageable memory: 0). */
int ClownResamplerAMD_DebugHostIsDeviceVisible(const void *host, size_t bytes);
/* The library at rest - to be called when no call is in progress on any thread: waits for every device it has used, then checks its
   process-wide state: no plan held by a call, every rows store referenced by exactly the plans that view it, every ticket block of every
   ring (and the unused part of the capture pool) reading zero, no lock held, every Debug* hook back at its default.  Returns the number
   of findings (0: at rest) and describes the first few in `message`. */
int ClownResamplerAMD_DebugSelfCheck(char *message, size_t capacity);
/* Flight recorder: the last 64 device operations of this library in this process - every kernel launch with the address ranges it was
   given (input + readable bytes, output + bytes, rows / table image, ticket block, stream, grid), every device allocation and release - written
   as text to the file descriptor `fd`; uses nothing but snprintf into a stack buffer and write(2).  With CLOWNRESAMPLER_AMD_ABORT_ON_ERROR=1 a
   failure writes it to stderr before it aborts.  ClownResamplerAMD_DebugInstallAbortDump has it written to stderr when the PROCESS receives SIGABRT - which
   is how the HIP runtime ends a process whose GPU reported a memory fault, from a thread of its own, some time after the faulting
   launch returned - and then passes the signal on to the handler that was installed before (0 on success). */
void ClownResamplerAMD_DebugDumpFlightRecorder(int fd);
int ClownResamplerAMD_DebugInstallAbortDump(void);

/* Debug/test access to the host copy of the polyphase rows (rows * row_stride int32). */
const int32_t *ClownResamplerAMD_PlanRows(const ClownResamplerAMD_Plan *plan);
/* Row index the kernels compute for a fractional position (host mirror of the device formula). */
uint32_t ClownResamplerAMD_PlanRowOf(const ClownResamplerAMD_Plan *plan, uint32_t position_fractional);

/* Test hook: route every launch through the generic 64-bit kernel (the independent second implementation). */
void ClownResamplerAMD_DebugForceGenericKernel(int on);
/* Tuning hook: selects the variant (geometry / arithmetic form) of the specialised kernels for plans created afterwards;
   the environment variable CLOWNRESAMPLER_AMD_VARIANT does the same at first use.  All variants give identical results. */
void ClownResamplerAMD_DebugSetVariant(int variant);
/* Diagnostic: device memory (32 bytes per workgroup) that the clock-stamp instance (variant 1006) fills with
   {shader cycles, start tick, end tick, XCC id} per workgroup (ticks of 10 ns). */
void ClownResamplerAMD_DebugSetStampBuffer(void *device_buffer);

/* device_input: interleaved int16, pointing at the start of the left padding, as for the reference call;
   at least *total_input_frames + 2 * integer_stretched_kernel_radius frames must be readable.
   device_output: int32, room for output_capacity_frames frames.  The launch is enqueued on hip_stream
   (NULL = the default stream) and NOT synchronised; the state and *total_input_frames are updated on the host
   from the closed form before returning.  Returns the number of frames that will have been written. */
size_t ClownResamplerAMD_ResampleDevice(ClownResamplerAMD_Plan *plan, ClownResampler_LowLevel_State *resampler, const void *device_input, size_t *total_input_frames, void *device_output, size_t output_capacity_frames, void *hip_stream, cc_bool *ran_out_of_input);

/* As ClownResamplerAMD_ResampleDevice, with device_output an int16 buffer and the clamp of
   ClownResampler_LowLevel_ResampleBulkS16 (4 bytes less write traffic per output sample). */
size_t ClownResamplerAMD_ResampleDeviceS16(ClownResamplerAMD_Plan *plan, ClownResampler_LowLevel_State *resampler, const void *device_input, size_t *total_input_frames, void *device_output, size_t output_capacity_frames, void *hip_stream, cc_bool *ran_out_of_input);

/* ---------------------------------------------------------------------------------------------
 * Variable rate (mid-stream ClownResampler_LowLevel_Adjust, clownresampler.h:1052-1056) on the device.
 * ------------------------------------------------------------------------------------------- */
typedef struct ClownResamplerAMD_Segment
{
	size_t input_frames;                    /* frames of the input timeline this segment covers */
	cc_u32f input_sample_rate;              /* arguments of the ClownResampler_LowLevel_Adjust applied before it */
	cc_u32f output_sample_rate;
	cc_u32f low_pass_filter_sample_rate;
} ClownResamplerAMD_Segment;

/* Resamples one contiguous device-resident timeline piecewise: for each segment, in order, the state is re-configured
   with ClownResampler_LowLevel_Adjust and the segment's input frames are resampled until they run out, the position
   (overshoot and fraction) carrying over - frame for frame what the reference produces when a caller alternates
   ClownResampler_LowLevel_Adjust and ClownResampler_LowLevel_Resample over consecutive chunks of one buffer.
   device_timeline points at input frame 0 (NOT at a padding); halo_frames frames must be readable before it and after
   the last segment's end, and every segment's integer_stretched_kernel_radius must fit in them (the caller zeroes the
   halo for the reference's zero padding at the ends of the stream).  device_output receives the segments' frames back
   to back: int32, or clamped int16 when output_is_s16.  segment_output_frames (may be NULL) receives the frame count
   of each segment.  One launch per non-empty segment on hip_stream, no synchronisation; everything is validated
   before the first launch (a rejected rate triple, a halo too small or an output too small is reported through the
   error handler, 0 is returned and nothing is enqueued).  On success *resampler is left as the reference leaves it
   after the last chunk, and the total frame count is returned. */
size_t ClownResamplerAMD_ResampleSegmentsDevice(ClownResampler_LowLevel_State *resampler, const ClownResampler_Precomputed *precomputed,
                                                const void *device_timeline, size_t halo_frames, const ClownResamplerAMD_Segment *segments, size_t segment_count,
                                                void *device_output, size_t output_capacity_frames, int output_is_s16, size_t *segment_output_frames, void *hip_stream);

/* ---------------------------------------------------------------------------------------------
 * Several GPUs, one process, one call (the reference has no counterpart).
 * The output timeline of ONE stream is split into shard_count contiguous blocks (ClownResamplerAMD_PlanShard) and block r
 * is computed on shards[r].device, on shards[r].hip_stream: an ordinary low-level call on that shard's slice of the
 * input whose padding is the real neighbouring frames (clownresampler.h:725-733).  There is no exchange between the
 * devices while they compute.  Optionally the blocks are then concatenated on one device, each transfer enqueued behind
 * its shard's kernel on that shard's stream:
 *   CLOWNRESAMPLER_AMD_GATHER_PEER_COPY  one hipMemcpyPeerAsync per shard (point-to-point over xGMI; exact sizes)
 *   CLOWNRESAMPLER_AMD_GATHER_RCCL       EXPERIMENTAL: grouped ncclSend / ncclRecv over a communicator set of the shards' devices, exact
 *                                        sizes like the copies (librccl.so is loaded on first use; every ordinal may appear only
 *                                        once - checked before anything is launched).  It has only ever run with ONE rank (the
 *                                        test pool has single-GPU boxes), where no NCCL operation is issued at all: treat it as
 *                                        unvalidated between distinct GPUs and prefer PEER_COPY until
 *                                        tests/test_gpu_ranks.py::test_two_distinct_gpus_over_rccl has run on a multi-GPU node.
 *                                        With MORE THAN ONE shard the call is refused (error ARGUMENT, nothing launched) unless
 *                                        CLOWNRESAMPLER_AMD_EXPERIMENTAL_RCCL=1 is set in the environment
 * Nothing is synchronised: use ClownResamplerAMD_ShardedSynchronize (or the streams) before reading.  Returns the total
 * number of output frames and leaves *resampler as ONE ClownResampler_LowLevel_Resample over the whole input would
 * (clownresampler.h:1065-1067); 0 after an error (reported through the handler).
 * ------------------------------------------------------------------------------------------- */
typedef struct ClownResamplerAMD_DeviceShard
{
	int device;                 /* HIP ordinal this shard runs on */
	const void *device_input;   /* ON that device: the shard's own padded slice, i.e. frame `first_input_frame` (ClownResamplerAMD_PlanShard)
	                               of the whole padded stream - its halo first; input_frames + 2 * halo_frames frames readable */
	void *device_output;        /* ON that device: room for the shard's output_frames (int32, or int16 with output_is_s16) */
	void *hip_stream;           /* a stream OF that device; NULL = its default stream */
} ClownResamplerAMD_DeviceShard;

enum
{
	CLOWNRESAMPLER_AMD_GATHER_NONE = 0,
	CLOWNRESAMPLER_AMD_GATHER_PEER_COPY = 1,
	CLOWNRESAMPLER_AMD_GATHER_RCCL = 2
};

size_t ClownResamplerAMD_ResampleShardedDevice(ClownResampler_LowLevel_State *resampler, const ClownResampler_Precomputed *precomputed, size_t total_input_frames,
                                               const ClownResamplerAMD_DeviceShard *shards, unsigned shard_count, int output_is_s16,
                                               int gather_mode, unsigned root_shard, void *root_output);
int ClownResamplerAMD_ShardedSynchronize(const ClownResamplerAMD_DeviceShard *shards, unsigned shard_count);

#ifdef __cplusplus
}
#endif

#endif /* CLOWNRESAMPLER_AMD_EXT_H */
