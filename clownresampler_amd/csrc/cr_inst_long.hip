// cr_inst_long.hip - instance unit: the 8-lobe build: 15- and 17-slot windows (BASELINE configs[2]: stereo 8 -> 96 kHz through k_up)  (see cr_instances.hpp)
#include "cr_instances.hpp"

namespace crk
{

int specials_long(void *table, int capacity)
{
	static const special mine[] = {
	    with_wave2<2, 15, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32, 16, 1, 4, 0x2A55u>(make_special<2, 15, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32, 27, true, false, 0x2A55u>()),  // cfg 3: stereo 8 -> 96 kHz, 8 lobes (k_up, chain form, from 2x upsampling on; k_wave below)
	    with_wave2<1, 15, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32, 16, 1, 4, 0x2A55u, true>(make_special_lite<1, 15, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32>()),
	    with_wave2<1, 17, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 16, 1, 4, 0u, true>(make_special_lite<1, 17, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_wave2<2, 17, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 16, 1, 2, 0u, true>(make_special_lite<2, 17, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),   // 8 lobes, stereo 48 -> 44.1 kHz: k_wave2 by default (144 against 171 us)
	};
	const int n = (int)(sizeof(mine) / sizeof(mine[0]));
	if (table == nullptr)
		return n;   // (asked for the count: specials() sizes its table from the providers)
	if (n > capacity)
		return -1;
	memcpy(table, mine, sizeof(mine));
	return n;
}

void *ablation_instance_long(int abl)
{
	switch (abl)
	{
		case 8: return (void *)k_up2<2, 15, CRHIP_NORM_U32, 0x2A55u, UP_WAVES, 0, 1, 6, 1>;    // k_up2 of the 8-lobe stereo instance + clock stamps
		case 9: return (void *)k_up2<2, 15, CRHIP_NORM_U32, 0x2A55u, UP_WAVES, 0, 1, 1, 1>;    // timing only: no global stores
		case 10: return (void *)k_up2<2, 15, CRHIP_NORM_U32, 0x2A55u, UP_WAVES, 0, 1, 2, 1>;   // timing only: no frames (window unpack + copy-out)
		case 11: return (void *)k_up2<2, 15, CRHIP_NORM_U32, 0x2A55u, UP_WAVES, 0, 1, 3, 1>;   // timing only: one row read per wave-tile instead of per frame
		case 12: return (void *)k_up2<2, 15, CRHIP_NORM_U32, 0x2A55u, UP_WAVES, 0, 1, 4, 1>;   // timing only: staging writes free of bank conflicts
		case 13: return (void *)k_up2<2, 15, CRHIP_NORM_U32, 0x2A55u, UP_WAVES, 0, 1, 5, 1>;   // timing only: both
		default: return nullptr;
	}
}

} // namespace crk
