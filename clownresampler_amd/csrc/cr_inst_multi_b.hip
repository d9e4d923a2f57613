// cr_inst_multi_b.hip - instance unit: 3 to 16 channels at 44.1 <-> 48 kHz, one instance each  (see cr_instances.hpp)
#include "cr_instances.hpp"

namespace crk
{

int specials_multi_b(void *table, int capacity)
{
	static const special mine[] = {
	    make_special_lite_chain<4, 5, CRHIP_NORM_S31, 0x12u>(),                          // quad, 5.1 and 7.1 at 44.1 <-> 48 kHz (upsampling: the 64-bit chain)
	    with_signed_chain<4, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 3>(make_special_lite<4, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    make_special_lite_chain<6, 5, CRHIP_NORM_S31, 0x12u>(),
	    with_signed_chain<6, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 4>(make_special_lite<6, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_signed_chain<8, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 2>(make_special_lite<8, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 2>()),            // (the geometry cfg 4's instance measured best with: 512 threads, plain stores; the chain with its 16 accumulator pairs: 374 against 241 us)
	    with_signed_chain<3, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 3>(make_special_lite<3, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31>()),               // and the odd layouts in between (2.1, 5.0, 6.1)
	    with_signed_chain<3, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 3>(make_special_lite<3, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    with_signed_chain<5, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 4>(make_special_lite<5, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31>()),
	    with_signed_chain<7, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, 4>(make_special_lite<7, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31>()),
	    with_signed_chain<7, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 4>(make_special_lite<7, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>()),
	    make_special_lite_split<12, 5, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31>(),        // 7.1.4 and 16 channels (the reference's maximum) at 44.1 <-> 48 kHz; (16,5), slower than the run-time instance with SDWA taps, came with the chain: cr_inst_multi_c.hip
	    make_special_lite_split<12, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
	    make_special_lite_split<16, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31>(),
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
