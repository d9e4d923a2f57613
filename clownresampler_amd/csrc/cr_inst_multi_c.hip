// cr_inst_multi_c.hip - instance unit: 9 to 16 channels at 44.1 <-> 48 kHz, two lanes per frame (odd counts: the last channel of the second lane a phantom);
// 12 channels and (16, 6) are in cr_inst_multi_b.hip  (see cr_instances.hpp)
#include "cr_instances.hpp"

namespace crk
{

int specials_multi_c(void *table, int capacity)
{
	static const special mine[] = {
	    make_special_lite_split_odd<9, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31>(),
	    make_special_lite_split_odd<9, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite_split_odd<11, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31>(),
	    make_special_lite_split_odd<11, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite_split_odd<13, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31>(),
	    make_special_lite_split_odd<13, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite_split_odd<15, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31>(),
	    make_special_lite_split_odd<15, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite_split<10, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31>(),
	    make_special_lite_split<10, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite_split<14, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31>(),
	    make_special_lite_split<14, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite_split<16, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31>(),
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
