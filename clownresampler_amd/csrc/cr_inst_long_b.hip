// cr_inst_long_b.hip - instance unit: the 8-lobe build at 44.1 <-> 48 kHz for 3 to 8 channels (15- and 17-slot windows): a
// specialised k_poly (the fallback) and k_wave2 (the default) each  (see cr_instances.hpp)
#include "cr_instances.hpp"

namespace crk
{

int specials_long_b(void *table, int capacity)
{
	// k_wave2 geometry: 16 waves, one 1 KiB window piece per wave-tile; 64 frames per wave-tile from 4 channels on (a frame of 4+
	// channels is 8+ bytes: 128 frames of window would not fit the piece), 128 below
	static const special mine[] = {
	    with_wave2<3, 15, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32, 16, 1, 2, 0x2A55u, true>(make_special_lite<3, 15, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32>()),
	    with_wave2<4, 15, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32, 16, 1, 1, 0x2A55u, true>(make_special_lite<4, 15, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32>()),
	    with_wave2<5, 15, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32, 16, 1, 1, 0x2A55u, true>(make_special_lite<5, 15, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32>()),
	    with_wave2<6, 15, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32, 16, 1, 1, 0x2A55u, true>(make_special_lite<6, 15, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32>()),
	    with_wave2<3, 17, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 16, 1, 2, 0u, true>(make_special_lite<3, 17, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_wave2<4, 17, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 16, 1, 1, 0u, true>(make_special_lite<4, 17, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_wave2<5, 17, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 16, 1, 1, 0u, true>(make_special_lite<5, 17, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    // windows of more than 1 KiB: two pieces per wave-tile, and as many waves as then fit beside the 66 KB of rows (pure
	    // upsampling: one configuration whatever the ratio, so the fit does not depend on it)
	    with_wave2<7, 15, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32, 11, 2, 1, 0x2A55u, true>(make_special_lite<7, 15, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32>()),
	    with_wave2<8, 15, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32, 11, 2, 1, 0x2A55u, true>(make_special_lite<8, 15, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32>()),
	    // 48 -> 44.1 kHz with 6 to 8 channels: k_wave2 with 10 waves measured 0.35 / 0.30 / 0.27 of the roofline against 0.29 / 0.24 / 0.27
	    // for the run-time-slot k_poly - but the specialised k_poly these entries would fall back to where a ratio's rows do not
	    // fit is far slower for 6 and 7 channels (0.19, 0.10: spills), so those two stay on the run-time-slot instance; for 8
	    // channels the specialised k_poly itself is the best of the three (0.32)
	    make_special_lite<8, 17, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),   // (the any-sign chain spills here: 17 slots x 8 channels, 118 -> 990 us)
	};
	const int n = (int)(sizeof(mine) / sizeof(mine[0]));
	if (table == nullptr)
		return n;   // (asked for the count: specials() sizes its table from the providers)
	if (n > capacity)
		return -1;
	memcpy(table, mine, sizeof(mine));
	return n;
}

} // namespace crk
