// cr_inst_seg.hip - instance unit: k_seg (cr_kseg.hpp), the segment-per-lane kernel of long stereo 8-lobe upsampling launches (BASELINE configs[2])
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "crhip.h"
#include "cr_kseg.hpp"

namespace
{
constexpr int SEG_WAVES = 12;
typedef void (*seg_fn)(const crhip_seg_launch);
// One shape of the one instance (stereo, 15 slots, pure upsampling): lines of 16 frames per lane and a ring of four groups - the ratios from 4x
// up, which advance at most once in four frames (4.5x ... 16x: 85-103 us per 40 M output frames where k_wave2 / k_up2 take 93-128,
// profiles/r05_seg_ratio_sweep.log).  The kernel is parameterised for more: with half-line chunks of 8 frames and a ring of eight groups it
// takes every ratio below 1:1 (bit-exact: tests ran it at 44.1 -> 48 kHz, 1.45x, 2.76x), but its half-line stores make it 149 us where
// k_wave2 takes 95 on the 8-lobe 44.1 -> 48 kHz workload, non-temporal or not (profiles/r05_kseg_hq48_first.log): not instantiated.
struct seg_shape
{
	seg_fn fn[5];          // [0] the kernel, [1 ... 3] timing-only ablations (results wrong), [4] cycle stamps per phase
	unsigned chunk, groups, lds;
};
// (the timing-only forms exist in a diagnostic build only - make CRA_CFLAGS=-DCRA_WITH_W2_FORMS - so that nothing in the environment can make
// a shipped library compute wrong samples; the stamped form computes what the kernel computes)
#ifdef CRA_WITH_W2_FORMS
#define CR_SEG_FORM(abl) k_seg<15, 0x2A55u, SEG_WAVES, 1, 16u, 4u, abl>
#else
#define CR_SEG_FORM(abl) nullptr
#endif
const seg_shape seg_shapes[1] = {
    {{k_seg<15, 0x2A55u, SEG_WAVES, 1, 16u, 4u>, CR_SEG_FORM(1), CR_SEG_FORM(2), CR_SEG_FORM(3), k_seg<15, 0x2A55u, SEG_WAVES, 1, 16u, 4u, 6>},
     16u, 4u, SEG_WAVES * seg_wave_bytes(16u, 4u) + 16u},
};
// the shape a ratio takes: the ring must hold what two chunks can advance over (k_seg, top_up) plus the group in use
const seg_shape *shape_of(uint32_t increment)
{
	for (const seg_shape &sh : seg_shapes)
		if (2u * ((65535u + sh.chunk * increment) >> 16) + 8u <= 4u * sh.groups)
			return &sh;
	return nullptr;
}
}

extern "C" {

int crhip_seg_instance(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode, uint32_t increment, uint32_t *negmask, uint32_t *threads, uint32_t *lds_bytes, uint32_t *chunk)
{
	const seg_shape *sh = shape_of(increment);
	if (channels != 2u || slots != 15u || row_mode != CRHIP_ROWMODE_UPSAMPLE || norm_mode != CRHIP_NORM_U32 || increment >= 65536u || sh == nullptr)
		return 0;
	*negmask = 0x2A55u;
	*threads = SEG_WAVES * 64u;
	*lds_bytes = sh->lds;
	*chunk = sh->chunk;
	return 1;
}

int crhip_seg_prepare(uint32_t channels, uint32_t slots, uint32_t increment, int *per_cu)
{
	const seg_shape *sh = shape_of(increment);
	if (channels != 2u || slots != 15u || sh == nullptr)
		return (int)hipErrorInvalidValue;
	hipError_t e = hipSuccess;
	for (int f = 0; f < 5 && e == hipSuccess; ++f)
		if (sh->fn[f] != nullptr)
			e = hipFuncSetAttribute((const void *)sh->fn[f], hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh->lds);
	if (e != hipSuccess)
		return (int)e;
	return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, (const void *)sh->fn[0], SEG_WAVES * 64, sh->lds);
}

int crhip_launch_seg(const crhip_seg_launch *launch, void *stream)
{
	const seg_shape *sh = shape_of(launch->increment);
	if (launch->slots != 15u || launch->blocks == 0 || sh == nullptr || launch->tile_frames % sh->chunk != 0 || launch->tiles_per_seg == 0
	 || (uint64_t)launch->tiles_per_seg * launch->tile_frames < launch->seg_frames || launch->n_tiles >= (1ull << 32))
		return (int)hipErrorInvalidValue;
	if (launch->n_out == 0)
		return 0;
	hipLaunchKernelGGL(sh->fn[(launch->debug_form < 5u && sh->fn[launch->debug_form] != nullptr) ? launch->debug_form : 0u], dim3(launch->blocks), dim3(SEG_WAVES * 64), sh->lds, (hipStream_t)stream, *launch);
	return (int)hipGetLastError();
}

} // extern "C"
