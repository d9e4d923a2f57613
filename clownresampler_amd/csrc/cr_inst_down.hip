// cr_inst_down.hip - instance unit: mono and stereo at the usual downsampling ratios (2:1, 96 -> 44.1, 3:2, 44.1 -> 32, 3:1, 44.1 -> 16, 44.1 -> 8, 88.2 -> 48)  (see cr_instances.hpp)
#include "cr_instances.hpp"

namespace crk
{

int specials_down(void *table, int capacity)
{
	static const special mine[] = {
	    with_signed_chain<1, 12, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 3>(make_special_lite<1, 12, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_wave2<2, 12, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 12, 2, 2, 0, true>(make_special_lite<2, 12, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_signed_chain<1, 13, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 3>(make_special_lite<1, 13, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_wave2<2, 13, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 16, 2, 2, 0, true>(make_special_lite<2, 13, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_signed_chain<1, 9, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 3>(make_special_lite<1, 9, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_signed_chain<2, 9, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 3>(make_special_lite<2, 9, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_signed_chain<1, 8, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 3>(make_special_lite<1, 8, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_signed_chain<2, 8, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 3>(make_special_lite<2, 8, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_signed_chain<1, 18, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 3>(make_special_lite<1, 18, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_wave2<1, 33, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 16, 2, 2, 0, true>(with_signed_chain<1, 33, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 3>(make_special_lite<1, 33, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>())),   // mono 44.1 -> 8 kHz: k_wave2 by default (68.8 -> 66.7 us, profiles/r04_dn8m_kwave2_ab.log; 256-frame wave-tiles do not fit beside the rows)
	    with_wave2<2, 33, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 16, 2, 1, 0, true>(make_special_lite<2, 33, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_wave2<2, 18, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 16, 2, 2, 0, true>(make_special_lite<2, 18, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    // high-rate material to CD / DAT rates: 176.4 -> 48 kHz (22 slots; also 44.1 -> 12 kHz), 192 -> 44.1 kHz (26 slots; 48 -> 11.025 kHz):
	    // stereo 0.40 -> 0.42 and 0.36 -> 0.41, mono 26 slots 0.25 -> 0.27 (mono 22 slots measured 3 % slower specialised: not instantiated)
	    with_wave2<2, 22, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 16, 2, 2, 0, true>(make_special_lite<2, 22, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_signed_chain<1, 26, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 3>(make_special_lite<1, 26, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_wave2<2, 26, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 16, 2, 1, 0, true>(make_special_lite<2, 26, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    // 44.1 -> 16 kHz (16 slots; the speech-recognition front end's conversion) and 88.2 -> 48 / 44.1 -> 24 kHz (11 slots)
	    with_signed_chain<1, 16, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 3>(make_special_lite<1, 16, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_wave2<2, 16, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 16, 2, 2, 0, true>(make_special_lite<2, 16, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_signed_chain<1, 11, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 3>(make_special_lite<1, 11, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_wave2<2, 11, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 16, 2, 2, 0, true>(make_special_lite<2, 11, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
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
