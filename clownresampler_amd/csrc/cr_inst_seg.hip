// cr_inst_seg.hip - instance unit: k_seg (cr_kseg.hpp), the segment-per-lane kernel of long stereo 8-lobe upsampling launches (BASELINE configs[2])
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "crhip.h"
#include "cr_kseg.hpp"

namespace
{
constexpr int SEG_WAVES = 12;
constexpr unsigned SEG_LDS = SEG_WAVES * SEG_WAVE_BYTES + 16u;
typedef void (*seg_fn)(const crhip_seg_launch);
const seg_fn seg_2_15 = k_seg<15, 0x2A55u, SEG_WAVES, 1>;
// diagnostic instances (crhip_seg_launch.debug_form): 1-3 timing-only ablations (results wrong), 5 = ten more scalar instructions per frame
const seg_fn seg_2_15_forms[7] = {seg_2_15, k_seg<15, 0x2A55u, SEG_WAVES, 1, 1>, k_seg<15, 0x2A55u, SEG_WAVES, 1, 2>, k_seg<15, 0x2A55u, SEG_WAVES, 1, 3>,
                                  seg_2_15, k_seg<15, 0x2A55u, SEG_WAVES, 1, 4>, k_seg<15, 0x2A55u, 16, 1, 0, 4u>};
// (form 6: sixteen waves - four per SIMD - with four future entries per tile: tiles of 32 frames at 12x)
constexpr unsigned SEG_LDS_16 = 16u * seg_wave_bytes(4u) + 16u;
}

extern "C" {

int crhip_seg_instance(uint32_t channels, uint32_t slots, uint32_t row_mode, uint32_t norm_mode, uint32_t *negmask, uint32_t *threads, uint32_t *lds_bytes)
{
	if (channels != 2u || slots != 15u || row_mode != CRHIP_ROWMODE_UPSAMPLE || norm_mode != CRHIP_NORM_U32)
		return 0;
	*negmask = 0x2A55u;
	*threads = SEG_WAVES * 64u;
	*lds_bytes = SEG_LDS;
	return 1;
}

int crhip_seg_prepare(uint32_t channels, uint32_t slots, int *per_cu)
{
	if (channels != 2u || slots != 15u)
		return (int)hipErrorInvalidValue;
	hipError_t e = hipSuccess;
	for (int f = 0; f < 7 && e == hipSuccess; ++f)
		e = hipFuncSetAttribute((const void *)seg_2_15_forms[f], hipFuncAttributeMaxDynamicSharedMemorySize, (int)(f == 6 ? SEG_LDS_16 : SEG_LDS));
	if (e != hipSuccess)
		return (int)e;
	return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, (const void *)seg_2_15, SEG_WAVES * 64, SEG_LDS);
}

int crhip_launch_seg(const crhip_seg_launch *launch, void *stream)
{
	if (launch->slots != 15u || launch->blocks == 0 || launch->tile_frames % 16u != 0 || launch->tiles_per_seg == 0
	 || (launch->tiles_per_seg & (launch->tiles_per_seg - 1u)) != 0)
		return (int)hipErrorInvalidValue;
	if (launch->n_out == 0)
		return 0;
	const unsigned form = launch->debug_form < 7u ? launch->debug_form : 0u;
	hipLaunchKernelGGL(seg_2_15_forms[form], dim3(launch->blocks), dim3(form == 6u ? 1024 : SEG_WAVES * 64), form == 6u ? SEG_LDS_16 : SEG_LDS, (hipStream_t)stream, *launch);
	return (int)hipGetLastError();
}

} // extern "C"
