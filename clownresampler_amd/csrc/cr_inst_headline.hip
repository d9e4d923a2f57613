// cr_inst_headline.hip - instance unit: mono / stereo with 3 lobes at 44.1 <-> 48 kHz (BASELINE configs[1], [4]) and the timing-only ablations of the headline instance  (see cr_instances.hpp)
#include "cr_instances.hpp"

namespace crk
{

int specials_headline(void *table, int capacity)
{
	static const special mine[] = {
	    make_special<2, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 28, true, true, 0x12u>(),   // cfg 2 / cfg 5: stereo 44.1 -> 48 kHz, 3 lobes: the 64-bit chain in its mov-armed form (variant 28; 13 = the SDWA form is its fallback)
	    with_chain<1, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 0x12u, 2>(make_special<1, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 28, true, true>()),   // mono upsampling, 3 lobes: two frames in flight per lane (79.9 -> 74.8 us on 20 minutes of mono)
	    with_signed_chain<2, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 3, true>(make_special<2, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 13, false, true>()),     // stereo mild downsampling, 3 lobes (+ the any-sign chain behind variants 28 / 29)
	    with_signed_chain<1, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 3, true, 2>(make_special<1, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 13, false, true>()),     // mono mild downsampling, 3 lobes: the any-sign chain with TWO frames in flight per lane (163.5 / 167.0 -> 160.5 / 161.0 us on 40 minutes of mono; with one frame in flight it measured slower than the SDWA form: 86.2 -> 89.5 us)
	};
	const int n = (int)(sizeof(mine) / sizeof(mine[0]));
	if (table == nullptr)
		return n;   // (asked for the count: specials() sizes its table from the providers)
	if (n > capacity)
		return -1;
	memcpy(table, mine, sizeof(mine));
	return n;
}

// timing-only ablations of the headline instance at the default geometry (see ABL in cr_kpoly.hpp)
void *ablation_instance(int abl)
{
	switch (abl)
	{
		case 1: return (void *)k_poly<2, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 1024, 1, 1, 1, 0, 1>;
		case 2: return (void *)k_poly<2, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 1024, 1, 1, 1, 0, 2>;
		case 3: return (void *)k_poly<2, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 1024, 1, 1, 1, 0, 3>;
		case 4: return (void *)k_poly<2, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 1024, 1, 1, 1, 0, 4>;
		case 5: return (void *)k_poly<2, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 1024, 1, 1, 1, 0, 4, 0, 1>;   // as 4, non-temporal stores
		case 6: return (void *)k_poly<2, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 1024, 1, 1, 1, 0, 6, 0, 1>;   // the real kernel (variant 13) + clock stamps
		case 7: return (void *)k_wave<2, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, WAVE_WAVES, WAVE_NVW, WAVE_ITER, 0, 1, 6>;   // k_wave + stamps
		case 8: case 9: case 10: case 11: case 12: case 13: return ablation_instance_long(abl);
		default: return nullptr;
	}
}

} // namespace crk
